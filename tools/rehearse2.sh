#!/bin/bash
# Runs on the GPU box: two bench.py ranks sharing GPU 0 over gloo, each under its own time limit,
# with Python tracebacks dumped if the limit hits (development aid for the N > 1 bench path).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/rehearse2; rm -rf "$OUT"; mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=2 PYTHONFAULTHANDLER=1
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 ${LIMIT:-60} python -u "$ROOT/bench.py" --gpus 2 --backend gloo --share-gpu --log2n ${LOG2N:-21} --steps ${STEPS:-2} --warmup ${WARM:-1} --no-cpu --no-extras > "$OUT/rank$r.out" 2> "$OUT/rank$r.err" &
done
wait
for r in 0 1; do echo "== rank $r"; tail -c 1500 "$OUT/rank$r.out"; grep -v "amdgpu.ids" "$OUT/rank$r.err" | tail -40; done
