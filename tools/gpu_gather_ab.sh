#!/bin/bash
# What the RCCL gather costs a rank (1-rank group on the one-GPU box, bench.py --force-dist): no gather / the bounded default /
# rarer, larger messages / smaller messages.  One JSON summary line per variant.
set -u
export TMPDIR=/tmp
O=gpurun_out/gather_ab; mkdir -p $O
COMMON="--force-dist --steps 40 --warmup 5 --no-cpu-baseline --no-single-step --no-other-configs --steady-launches 0"
run() { local name=$1; shift; timeout -k 10 200 python3 bench.py $COMMON "$@" > $O/$name.json 2> $O/$name.err; python3 - "$name" "$O/$name.json" <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        b = json.loads(l)
        print(sys.argv[1], "ms/step %.4f" % b["ms_per_step"], "kernel_ms %.4f" % b["roofline"]["kernel_ms"], "value %.3e" % b["value"],
              {k: b["config"].get(k) for k in ("gather", "gather_steps_per_message", "gather_every_chunks")})
PY
}
run none --gather none
run bounded_every4
run bounded_every8 --gather-every 8
run bounded_every1 --gather-every 1
run bounded_every16 --gather-every 16
