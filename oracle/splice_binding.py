#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes to STDOUT the reference's swipe.cc with search_chunk() (swipe.cc:1365-1596) replaced by
the binding of INTEGRATION.md section 2 - the prelude block plus binding A ("scores") or B ("topk"), cut out of the
document, so the document IS the code that runs.  The patched text only ever exists in the pipe to the compiler
(oracle/Makefile); no copy of a reference source is written anywhere.

usage: splice_binding.py /root/reference/swipe.cc INTEGRATION.md scores|topk|group   (group = topk, built with -DSWIPE_AMD_GROUP)"""
import re
import sys


def blocks(doc):
    return re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)


def binding(doc, variant):
    guard = {"scores": "#ifdef SWIPE_AMD_SCORES", "topk": "#ifdef SWIPE_AMD_TOPK", "group": "#ifdef SWIPE_AMD_TOPK"}[variant]
    b = blocks(doc)
    prelude = next(x for x in b if x.startswith("#ifdef SWIPE_AMD\n"))
    body = next(x for x in b if x.startswith(guard))
    return prelude + body


def splice(src, code):
    start = src.index("void search_chunk(struct search_data * sdp)\n{")
    end = src.index("void * worker(void *)")
    return src[:start] + "#ifndef SWIPE_AMD\n" + src[start:end] + "#endif\n" + code + "\n" + src[end:]


if __name__ == "__main__":
    ref, doc, variant = sys.argv[1:4]
    sys.stdout.write(splice(open(ref).read(), binding(open(doc).read(), variant)))
