#!/bin/bash
# round-3 measurement batch (run on the GPU box through gpurun): A/B of kernel builds, phase stamps, LDS PMC pass, MARL baselines, tests
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
for r in 1 2; do for p in r2 p5 p7; do echo -n "$p " >> $O/ab_bench.txt; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --lib mop-truss-marl_amd/csrc/abl/libtruss_$p.so 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rollout_us', round(d['roofline']['kernel_us'],2), 'step_us', round(d['one_launch_per_step']['kernel_us'],2), 'fused_us', round(d['state_emitting_step']['us_per_step'],2))" >> $O/ab_bench.txt || exit 1; done; done
python tools/phase_stamps.py 4096 > $O/phase.txt 2>&1
python tools/fused_stamps.py 4096 mop-truss-marl_amd/csrc/libtruss_mi355_diag.so all 2>&1 | grep -v amdgpu.ids > $O/stamps_all.txt
for c in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  tag=$(echo $c | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $c -d $O/pmc_$tag -o pmc -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2> $O/pmc_$tag.err || { echo "pmc pass $c failed" >> $O/pmc.txt; tail -3 $O/pmc_$tag.err >> $O/pmc.txt; }
done
python tools/reduce_profile.py $O 20 > $O/pmc_summary.json 2>> $O/pmc.txt
timeout -k 10 300 python tools/marl_bench.py 4096 4 0 > $O/marl_notrain.txt 2>&1
timeout -k 10 300 python tools/marl_bench.py 4096 4 1 > $O/marl_train.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
find $O -name "*.db" -size +8M -delete; find $O -name "*.csv" -size +2M -delete
cat $O/ab_bench.txt $O/phase.txt $O/stamps_all.txt; tail -3 $O/marl_notrain.txt $O/marl_train.txt $O/pytest.txt; head -c 1500 $O/pmc_summary.json
