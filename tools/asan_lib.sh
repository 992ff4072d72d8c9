#!/bin/bash
# libswipe_amd.so's HOST side under AddressSanitizer + UBSan inside the Python tests (make -C swipe_amd/csrc asan builds
# swipe_amd/libswipe_amd_asan.so: every host translation unit instrumented, the kernel objects of the normal build linked
# as they are).  The instrumented library takes the place of libswipe_amd.so and the sanitizer runtime is preloaded into
# the interpreter.
#   bash tools/asan_lib.sh            CPU tests of the host logic (reader, headers, statistics, traceback, merges, shard
#                                     bounds, kernel choice, option table): runs without a GPU, in the build container
#   bash tools/asan_lib.sh gpu [s]    the host-heavy GPU parity tests.  On this image the PRELOADED runtime's
#                                     hsa_amd_memory_pool_allocate interceptor fails inside the HIP runtime ("allocator is
#                                     trying to allocate 0x400000 bytes", profiles/r03_asan.txt), so on a GPU box the
#                                     statically linked driver of tools/asan_cli.sh is what runs instrumented.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
test -f swipe_amd/libswipe_amd_asan.so || { echo "no swipe_amd/libswipe_amd_asan.so: make -C swipe_amd/csrc asan"; exit 2; }
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p gpurun_out/asan
cp swipe_amd/libswipe_amd.so swipe_amd/libswipe_amd.so.plain
trap 'cp swipe_amd/libswipe_amd.so.plain swipe_amd/libswipe_amd.so' EXIT
cp swipe_amd/libswipe_amd_asan.so swipe_amd/libswipe_amd.so
export ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1
if [ "$1" = "gpu" ]; then
  K="scores_equal_reference or hit_list_equals or open_blast_volumes or width_escalation or requeue_by_batches or bound_build_is_dropped"
  K="$K or alignment_end_points or empty_inputs or permissive_threshold or dual_query_kernel_both or hand_over_buffer or alignment_phase_matches"
  K="$K or translated_scores or six_frame or translated_hit_list or inclusion_subset or options_are_explicit or requeue_list_longer or 64_bit_hop"
  K="$K or residue_codes or long_subject or windows_are_shorter or titin or span_long_gaps or two_different_queries or hbm_budget or windows_compose or seeded_fuzz"
  LD_PRELOAD=$RT timeout ${2:-600} python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "$K" 2>&1 | tail -25
else
  # tests that compile or start other programs are left out (the preload would follow them)
  LD_PRELOAD=$RT timeout 900 python -m pytest tests/test_host_cpu.py tests/test_oracle_golden.py -m "not gpu" -q -p no:cacheprovider \
    -k "not sanitizer and not damaged and not rescoring and not register_room and not integration and not binding" 2>&1 | tail -8
fi
rc=${PIPESTATUS[0]}
echo "pytest rc=$rc"
exit $rc
