#!/bin/bash
# Run a command with the product library's HIP calls served by the gfx950 interpreter:   tools/gfx950sim/run.sh python -m pytest tests -m gpu -k ...
D="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
make -s -C "$D" libhipsim.so || exit 1
export HIPSIM=1
exec env LD_PRELOAD="$D/libhipsim.so${LD_PRELOAD:+:$LD_PRELOAD}" "$@"
