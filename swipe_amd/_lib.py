"""ctypes binding of swipe_amd/libswipe_amd.so (the C ABI in include/swipe_amd.h).

The library is the product: if it is missing this module raises - there is no Python or CPU
fallback for any compute entry point."""
from __future__ import annotations

import ctypes as C
import os

SWA_ERANGE = -6

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWA_LIB: another build of the same library (tools/device_asan.sh: the kernels under AddressSanitizer); never a fallback
LIB_PATH = os.environ.get("SWA_LIB") or os.path.join(_HERE, "libswipe_amd.so")


class DbInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("seqcount", "symcount", "longest", "first_seqno",
                                         "total_seqcount", "total_symcount", "hbm_bytes", "frames")]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("narrow", "wide", "full", "cells")] + \
               [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("narrow_rows", C.c_int32),
                ("narrow_shifted", C.c_int32), ("loading_parts", C.c_int32), ("requeue_form", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("seqno", C.c_int64), ("score", C.c_int64)]


class Alignment(C.Structure):
    _fields_ = [("seqno", C.c_int64), ("dstrand", C.c_int32), ("dframe", C.c_int32), ("hinted", C.c_int32),
                ("reserved", C.c_int32)] + \
               [(n, C.c_int64) for n in ("score", "q_start", "q_end", "d_start", "d_end", "dlen", "dlennt", "identities",
                                         "positives", "indels", "aligned", "gaps", "cigar_offset", "cigar_len")]


class FrameHit(C.Structure):
    _fields_ = [("seqno", C.c_int64), ("score", C.c_int64)] + \
               [(n, C.c_int32) for n in ("qstrand", "qframe", "dstrand", "dframe")]


class Stats(C.Structure):
    _fields_ = [("available", C.c_int)] + \
               [(n, C.c_double) for n in ("lam", "K", "H", "alpha", "beta", "Kmn", "logK",
                                          "lambda_d_log2", "logK_d_log2")] + \
               [(n, C.c_int64) for n in ("lenadj", "m", "n", "scorethreshold", "upperscorethreshold")]


EXPORTS = [
    "swa_last_error", "swa_device_count", "swa_redzones_check", "swa_db_open", "swa_db_open_async", "swa_db_wait", "swa_db_load_progress", "swa_debug_load_layout", "swa_db_from_memory", "swa_db_from_memory_streamed", "swa_db_open_streamed", "swa_db_info",
    "swa_db_open_translated", "swa_db_from_memory_translated", "swa_search_frames_topk",
    "swa_gencode_name", "swa_translate_table", "swa_translate",
    "swa_headers_open", "swa_headers_close", "swa_headers_info", "swa_headers_time", "swa_headers_get", "swa_headers_inclusion",
    "swa_db_set_inclusion", "swa_set_option",
    "swa_db_close", "swa_blastdb_read", "swa_blastdb_write", "swa_free", "swa_blastdb_defline", "swa_blastdb_deflines", "swa_set_scoring", "swa_search", "swa_search_topk", "swa_search2", "swa_search2_topk", "swa_search_pair_topk", "swa_search_endpoints", "swa_search_endpoints_strand",
    "swa_db_sequence", "swa_align_hits", "swa_traceback", "swa_hits_merge", "swa_fhits_merge",
    "swa_stats_init", "swa_evalue", "swa_bits", "swa_matrix_builtin", "swa_matrix_nucleotide",
    "swa_matrix_parse", "swa_default_gaps",
    "swa_synth_length", "swa_synth_offsets", "swa_synth_fill",
    "swa_shard_bounds", "swa_blastdb_shard_bounds", "swa_group_open", "swa_group_open_streamed", "swa_group_from_memory", "swa_group_close", "swa_group_info",
    "swa_group_shard", "swa_group_wait", "swa_group_load_progress", "swa_group_set_scoring", "swa_group_set_option", "swa_group_set_inclusion", "swa_group_search",
    "swa_group_search_topk", "swa_group_search_pair_topk", "swa_group_search_frames_topk", "swa_group_align_hits",
    "swa_group_db_sequence", "swa_kernel_choice", "swa_kernel_rate", "swa_kernel_choice2", "swa_kernel_rate2",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C swipe_amd/csrc` (swipe_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i64, i64p = C.c_void_p, C.c_int64, C.POINTER(C.c_int64)
    L.swa_last_error.restype = C.c_char_p
    L.swa_device_count.restype = C.c_int
    L.swa_db_open.argtypes = [C.c_char_p, C.c_int, C.c_int, i64, i64, C.POINTER(vp)]
    L.swa_redzones_check.argtypes = [i64p, i64p, C.c_char_p, i64]
    L.swa_db_open_async.argtypes = L.swa_db_open.argtypes
    L.swa_db_wait.argtypes = [vp]
    L.swa_debug_load_layout.argtypes = [vp, i64, i64, vp]
    L.swa_db_load_progress.argtypes = [vp, i64p, i64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.swa_db_from_memory.argtypes = [vp, vp, i64, C.c_int, C.c_int, i64, i64, i64, C.POINTER(vp)]
    L.swa_db_from_memory_streamed.argtypes = [vp, vp, i64, C.c_int, C.c_int, i64, i64, i64, i64, C.POINTER(vp)]
    L.swa_db_open_streamed.argtypes = [C.c_char_p, C.c_int, C.c_int, i64, i64, i64, C.POINTER(vp)]
    L.swa_db_info.argtypes = [vp, C.POINTER(DbInfo)]
    L.swa_blastdb_read.argtypes = [C.c_char_p, C.c_int, i64, i64, C.POINTER(vp), C.POINTER(vp), i64p, i64p, i64p, i64p]
    L.swa_blastdb_defline.argtypes = [C.c_char_p, C.c_int, i64, C.c_char_p, i64, i64p]
    L.swa_blastdb_deflines.argtypes = [C.c_char_p, C.c_int, i64, C.c_char_p, i64, i64p]
    L.swa_blastdb_write.argtypes = [C.c_char_p, C.c_int, vp, vp, i64, i64, C.c_char_p]
    L.swa_free.argtypes = [vp]
    L.swa_free.restype = None
    L.swa_db_close.argtypes = [vp]
    L.swa_db_close.restype = None
    L.swa_set_scoring.argtypes = [vp, vp, i64, i64]
    L.swa_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.swa_search.argtypes = [vp, vp, i64, vp, C.POINTER(Counters)]
    L.swa_search_topk.argtypes = [vp, vp, i64, i64, i64, i64, C.POINTER(Hit), i64p, i64p, i64p, C.POINTER(Counters)]
    L.swa_search2.argtypes = [vp, vp, vp, i64, vp, vp, C.POINTER(Counters)]
    L.swa_search2_topk.argtypes = [vp, vp, vp, i64, i64, i64, i64, C.POINTER(Hit), C.POINTER(C.c_int32), i64p, i64p, i64p,
                                   C.POINTER(Counters)]
    L.swa_search_pair_topk.argtypes = [vp, vp, i64, vp, i64, i64, i64, i64, i64, i64, i64, C.POINTER(Hit), i64p, i64p, i64p,
                                       C.POINTER(Hit), i64p, i64p, i64p, C.POINTER(Counters)]
    L.swa_search_endpoints.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp]
    L.swa_search_endpoints_strand.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, vp, vp]
    L.swa_db_sequence.argtypes = [vp, i64, C.c_int, C.c_int, vp, i64, i64p, i64p]
    L.swa_align_hits.argtypes = [vp, vp, i64, vp, vp, vp, i64, C.POINTER(Alignment), C.c_char_p, i64, i64p]
    L.swa_db_open_translated.argtypes = [C.c_char_p, C.c_int, C.c_int, i64, i64, C.POINTER(vp)]
    L.swa_db_from_memory_translated.argtypes = [vp, vp, i64, C.c_int, C.c_int, i64, i64, i64, C.POINTER(vp)]
    L.swa_search_frames_topk.argtypes = [vp, C.c_int, C.POINTER(vp), i64p, C.POINTER(C.c_int32), i64, i64, i64,
                                         C.POINTER(FrameHit), i64p, i64p, i64p, C.POINTER(Counters)]
    L.swa_headers_open.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.POINTER(vp)]
    L.swa_headers_close.argtypes = [vp]
    L.swa_headers_close.restype = None
    L.swa_headers_info.argtypes = [vp, i64p, i64p, i64p, i64p, i64p, C.c_char_p, i64]
    L.swa_headers_time.argtypes = [vp, C.c_char_p, i64]
    L.swa_headers_get.argtypes = [vp, i64, C.c_int, C.c_char_p, i64, i64p]
    L.swa_headers_inclusion.argtypes = [vp, i64, i64, vp]
    L.swa_db_set_inclusion.argtypes = [vp, vp, i64]
    L.swa_gencode_name.argtypes = [C.c_int]
    L.swa_gencode_name.restype = C.c_char_p
    L.swa_translate_table.argtypes = [C.c_int, vp]
    L.swa_translate.argtypes = [vp, i64, C.c_int, C.c_int, vp, vp, i64p]
    L.swa_traceback.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, i64, C.POINTER(Alignment), C.c_char_p, i64, i64p]
    L.swa_hits_merge.argtypes = [C.POINTER(Hit), i64p, C.c_int, i64, i64, C.POINTER(Hit), i64p]
    L.swa_fhits_merge.argtypes = [C.POINTER(FrameHit), i64p, C.c_int, i64, i64, C.POINTER(FrameHit), i64p]
    L.swa_stats_init.argtypes = [C.c_int, C.c_char_p, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64,
                                 C.c_double, C.c_double, C.POINTER(Stats)]
    L.swa_evalue.argtypes = [C.POINTER(Stats), i64]
    L.swa_evalue.restype = C.c_double
    L.swa_bits.argtypes = [C.POINTER(Stats), i64]
    L.swa_bits.restype = C.c_double
    L.swa_matrix_builtin.argtypes = [C.c_char_p, vp]
    L.swa_matrix_nucleotide.argtypes = [i64, i64, vp]
    L.swa_matrix_parse.argtypes = [C.c_char_p, vp]
    L.swa_default_gaps.argtypes = [C.c_char_p, i64p, i64p]
    L.swa_synth_length.argtypes = [C.c_uint64, i64, vp, vp, i64]
    L.swa_synth_length.restype = i64
    L.swa_synth_offsets.argtypes = [C.c_uint64, i64, i64, vp, vp, i64, vp, C.c_int]
    L.swa_synth_offsets.restype = i64
    L.swa_synth_fill.argtypes = [C.c_uint64, i64, i64, vp, vp, vp, i64, vp, vp, C.c_int]
    ip = C.POINTER(C.c_int)
    L.swa_shard_bounds.argtypes = [vp, i64, C.c_int, vp]
    L.swa_blastdb_shard_bounds.argtypes = [C.c_char_p, C.c_int, C.c_int, vp]
    L.swa_group_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, ip, C.POINTER(vp)]
    L.swa_group_open_streamed.argtypes = [C.c_char_p, C.c_int, C.c_int, ip, i64, C.POINTER(vp)]
    L.swa_group_from_memory.argtypes = [vp, vp, i64, C.c_int, C.c_int, C.c_int, ip, i64, i64, i64, C.POINTER(vp)]
    L.swa_group_close.argtypes = [vp]
    L.swa_group_close.restype = None
    L.swa_group_info.argtypes = [vp, C.POINTER(DbInfo), ip]
    L.swa_group_shard.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.swa_group_set_scoring.argtypes = [vp, vp, i64, i64]
    L.swa_group_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.swa_group_wait.argtypes = [vp]
    L.swa_group_load_progress.argtypes = [vp, i64p, i64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.swa_group_set_inclusion.argtypes = [vp, vp, i64]
    L.swa_group_search.argtypes = L.swa_search.argtypes
    L.swa_group_search_topk.argtypes = L.swa_search_topk.argtypes
    L.swa_group_search_pair_topk.argtypes = L.swa_search_pair_topk.argtypes
    L.swa_group_search_frames_topk.argtypes = L.swa_search_frames_topk.argtypes
    L.swa_group_align_hits.argtypes = L.swa_align_hits.argtypes
    L.swa_group_db_sequence.argtypes = L.swa_db_sequence.argtypes
    L.swa_kernel_choice.argtypes = [i64, C.c_int, i64, i64, i64, i64, C.c_double, C.c_int] + [C.POINTER(C.c_int32)] * 4
    L.swa_kernel_rate.argtypes = [C.c_int, C.c_int, C.c_int]
    L.swa_kernel_choice2.argtypes = [C.c_int, i64, C.c_int, i64, i64, i64, i64, C.c_double, C.c_int] + [C.POINTER(C.c_int32)] * 4
    L.swa_kernel_rate2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L
