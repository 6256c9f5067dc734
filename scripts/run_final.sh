#!/bin/bash
# end-of-round validation on the GPU box: full GPU test suite, contract bench, training benches.  bash scripts/run_final.sh <tag>
tag=${1:-r02z}
out=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/${tag}_tests.log 2>&1; echo rc=$? >> $out/${tag}_tests.log
grep -E "passed|failed|FAILED|^rc=" $out/${tag}_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
python bench.py --config baseline_sh --no-cpu-baseline > $out/${tag}_bench_sh.json 2>> $out/${tag}_bench.err || exit 1
python bench.py --config config5 --seconds 30 --triplets 8 --no-cpu-baseline > $out/${tag}_bench_c5.json 2>> $out/${tag}_bench.err || exit 1
python bench.py --aug --no-cpu-baseline > $out/${tag}_bench_aug.json 2>> $out/${tag}_bench.err || exit 1
python bench.py --ingest pcm16 --no-cpu-baseline > $out/${tag}_bench_ingest_pcm16.json 2>> $out/${tag}_bench.err || exit 1
python bench.py --ingest f32 --no-cpu-baseline > $out/${tag}_bench_ingest_f32.json 2>> $out/${tag}_bench.err || exit 1
for p in fp32 f16x3 f16 amp; do python bench.py --train --train-precision $p > $out/${tag}_train_$p.json 2>> $out/${tag}_bench.err || exit 1; done
for p in fp32 f16x3 f16; do python bench.py --train --train-precision $p --config baseline_sh > $out/${tag}_train_sh_$p.json 2>> $out/${tag}_bench.err || exit 1; done
for p in fp32 f16x3 f16; do python bench.py --train --train-precision $p --config config5 --seconds 30 --triplets 8 > $out/${tag}_train_c5_$p.json 2>> $out/${tag}_bench.err || exit 1; done
python - "$tag" <<'PY'
import json, glob, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/{tag}_bench.json"))
print("contract", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("stage_a_ms"), d["roofline"].get("stage_a_hbm_frac"))
for f in sorted(glob.glob(f"gpurun_out/{tag}_bench_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"].get("stage_a_ms"), d["roofline"].get("kernels_ms"))
for f in sorted(glob.glob(f"gpurun_out/{tag}_train*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["loss"], d["config"]["peak_mem_GiB"])
PY
