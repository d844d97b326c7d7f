#!/bin/bash
# Re-measures every configuration of the project's measurement table on one MI355X (a few minutes in total).
# usage: tools/reproduce.sh            results: gpurun_out/repro_<tag>.json, one bench.py line each
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out"
run() {   # tag, bench args...
  local tag=$1; shift
  timeout -k 10 600 python3 "$ROOT/bench.py" --no-cpu-baseline --no-also "$@" > "$ROOT/gpurun_out/repro_$tag.log" 2>&1 || { echo "$tag FAILED"; tail -5 "$ROOT/gpurun_out/repro_$tag.log"; return 1; }
  grep "^{" "$ROOT/gpurun_out/repro_$tag.log" > "$ROOT/gpurun_out/repro_$tag.json"
  python3 - "$tag" "$ROOT/gpurun_out/repro_$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())
r = d["roofline"]
print("%-22s %8.2f ms/iter  %.3g nnz/s  %5.0f GB/s algorithmic (%.2f of HBM peak)  %s" % (
    sys.argv[1], d["ms_per_step"], d["value"], r["achieved"], r["frac"], r["launch_ms"]), flush=True)
PY
}
run c2_f64_exact                                   && \
run c2_f64_exact_bf16x3  --accumulate bf16x3       && \
run c2_f64_exact_fp32    --accumulate fp32         && \
run c2_f64_cgdef         --solver cg               && \
run c2_f64_cg3           --solver cg --cg-steps 3  && \
run c2_f128_exact        --factors 128 --steps 5   && \
run c2_f128_exact_bf16x3 --factors 128 --steps 5 --accumulate bf16x3 && \
run c2_f128_exact_fp32   --factors 128 --steps 5 --accumulate fp32 && \
run c2_f128_exact_wave   --factors 128 --steps 5 --row-kernel wave && \
run c2_f128_cgdef        --factors 128 --steps 5 --solver cg && \
run c2_f128_cg3          --factors 128 --steps 5 --solver cg --cg-steps 3 && \
run c2_f128_cg3_lockstep --factors 128 --steps 5 --solver cg --cg-steps 3 --cg-form lock_step && \
run c2_f256_cgdef        --factors 256 --steps 3 --solver cg && \
run c2_f256_cg3          --factors 256 --steps 5 --solver cg --cg-steps 3 && \
run c3shard_exact        --workload c3_shard_syn --steps 3 --warmup 1 && \
run c3shard_exact_bf16x3 --workload c3_shard_syn --steps 3 --warmup 1 --accumulate bf16x3 && \
run c3shard_exact_noshort --workload c3_shard_syn --steps 3 --warmup 1 --short-rows off && \
run c3shard_exact_bf16x3_noshort --workload c3_shard_syn --steps 3 --warmup 1 --accumulate bf16x3 --short-rows off && \
run c3shard_cgdef        --workload c3_shard_syn --steps 3 --warmup 1 --solver cg && \
run c3shard_cg3          --workload c3_shard_syn --steps 3 --warmup 1 --solver cg --cg-steps 3 && \
run c3rank_exact         --workload c3_rank_syn --steps 3 --warmup 1 && \
run c3rank_exact_bf16x3  --workload c3_rank_syn --steps 3 --warmup 1 --accumulate bf16x3 && \
run c3rank_exact_noshort --workload c3_rank_syn --steps 3 --warmup 1 --short-rows off && \
run c3rank_cgdef         --workload c3_rank_syn --steps 3 --warmup 1 --solver cg && \
run c3full_exact         --workload c3_full_syn --steps 2 --warmup 1 && \
run c4shard_cgdef        --workload c4_shard_syn --steps 2 --warmup 1 && \
run c4shard_cg3          --workload c4_shard_syn --steps 2 --warmup 1 --cg-steps 3 && \
python3 "$ROOT/tools/bench_topk.py" --users 65536 --skip-materialised && \
python3 "$ROOT/tools/bench_topk.py" --users 65536 --skip-materialised --prepared && \
python3 "$ROOT/tools/bench_topk.py" --users 1048576 --filter-rated --skip-materialised && \
python3 "$ROOT/tools/bench_topk.py" --users 1048576 --filter-rated --skip-materialised --prepared && python3 "$ROOT/tools/bench_gram.py"
