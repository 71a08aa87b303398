#!/bin/bash
# Runs ON THE GPU BOX: a long pass of all the randomised sweeps with fresh seeds (bash tools/fuzz_all.sh <first seed>); stops at the
# first run that ends with anything but "no mismatch" / "mismatch" (a crash), prints one summary line per run.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
s0=${1:-100}
run() {  # name, env, script, args...
  name=$1; envs=$2; shift 2
  env $envs timeout -k 10 280 python3 -u "$@" > $out/fa_$name.log 2>&1
  r=$?
  echo "$name rc=$r: $(tail -1 $out/fa_$name.log | cut -c1-200)"
  grep -B3 -A10 MISMATCH $out/fa_$name.log | grep -v '^  File\|^Traceback' | cut -c1-300 | head -30
  [ $r -le 1 ] || { tail -8 $out/fa_$name.log; exit $r; }
}
for k in 0 1 2; do run pairs$k X=1 $root/tools/fuzz_parity.py 400 $((s0 + k)); done
for k in 0 1; do run extreme$k FUZZ_EXTREME=1 $root/tools/fuzz_parity.py 300 $((s0 + 10 + k)); done
run big0 FUZZ_BIG=1 $root/tools/fuzz_parity.py 80 $((s0 + 20))
for k in 0 1; do run streams$k X=1 $root/tools/fuzz_streams.py 100 $((s0 + 30 + k)); done
run mix0 X=1 $root/tools/fuzz_api_mix.py 60 $((s0 + 40))
run mixbig0 FUZZ_MIX_BIG=1 $root/tools/fuzz_api_mix.py 30 $((s0 + 41))
