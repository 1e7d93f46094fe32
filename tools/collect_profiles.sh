#!/bin/bash
# Round artefacts on the MI355X box: bench lines of every workload, rocprofv3 kernel stats, PMC traffic passes, GEMM / inverse probes.
# usage (through gpurun, from the repo root): bash tools/collect_profiles.sh r02a <commit>     -> files under gpurun_out/r02a_*
# (copy what is to be judged into profiles/ afterwards; see profiles/README.md)
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
python bench.py --workload cfg3 > $O/${tag}_cfg3.json 2>> $O/${tag}_bench.err
python bench.py --workload sprites800 > $O/${tag}_sprites800_f64.json 2>> $O/${tag}_bench.err
python bench.py --workload sprites800 --precision f32 > $O/${tag}_sprites800_f32.json 2>> $O/${tag}_bench.err
python bench.py --workload cfg5 > $O/${tag}_cfg5.json 2>> $O/${tag}_bench.err
python bench.py --workload sprites800 --precision f32 --kernel se > $O/${tag}_sprites800_f32_se.json 2>> $O/${tag}_bench.err
# the self-launching multi-rank path with one rank: strong + weak lines through torch.distributed.run spawned by bench.py itself
python bench.py --gpus 1 --force-dist --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_cfg2_self_launch.json 2>> $O/${tag}_bench.err
python bench.py --workload cfg3 --force-comm --no-cpu-baseline > $O/${tag}_cfg3_force_comm.json 2>> $O/${tag}_bench.err
python bench.py --force-comm --no-cpu-baseline > $O/${tag}_cfg2_force_comm.json 2>> $O/${tag}_bench.err
python bench.py --workload sprites800 --precision f32 --force-comm --no-cpu-baseline > $O/${tag}_sprites800_f32_force_comm.json 2>> $O/${tag}_bench.err
python tools/gemm_sweep.py 2>/dev/null > $O/${tag}_gemm_sweep.txt
python tools/inverse_probe.py 2>/dev/null > $O/${tag}_inverse_probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_cfg2 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_prof_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_cfg3 -- python3 $R/bench.py --workload cfg3 --steps 50 --warmup 10 --repeats 1 --no-cpu-baseline > $O/${tag}_prof_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_sp800 -- python3 $R/bench.py --workload sprites800 --precision f32 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/${tag}_prof_sp800.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_cfg5 -- python3 $R/bench.py --workload cfg5 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > $O/${tag}_prof_cfg5.log 2>&1
export SVGP_BENCH_NO_STAGES=1      # counter passes: the step only, no per-stage timing launches
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/${tag}_pmc_$c/cfg2 -- python3 $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline > $O/${tag}_pmc_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/${tag}_pmc_$c/cfg3 -- python3 $R/bench.py --workload cfg3 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline >> $O/${tag}_pmc_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/${tag}_pmc_$c/sp800 -- python3 $R/bench.py --workload sprites800 --precision f32 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline >> $O/${tag}_pmc_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/${tag}_pmc_$c/cfg5 -- python3 $R/bench.py --workload cfg5 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline >> $O/${tag}_pmc_$c.log 2>&1
done
# matrix-pipe busy / wave wait shares (one SQ pass): the GEMM, convolution, Cholesky and Gauss-Jordan kernels in their steps
for w in "cfg3:--workload cfg3 --steps 10 --warmup 2" "sp800:--workload sprites800 --precision f32 --steps 2 --warmup 1"; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${tag}_pmc_sq/${w%%:*} -- python3 $R/bench.py ${w#*:} --repeats 1 --no-cpu-baseline > $O/${tag}_pmc_sq_${w%%:*}.log 2>&1
done
unset SVGP_BENCH_NO_STAGES
cd $R
python tools/pmc_summary.py $O/${tag}_pmc_FETCH_SIZE $O/${tag}_pmc_WRITE_SIZE $O/${tag}_pmc_traffic.json
python tools/mfma_busy_summary.py $O/${tag}_pmc_sq $O/${tag}_mfma_busy.json
python tools/conv_probe.py 500 f32 > $O/${tag}_conv_probe_f32.txt 2>/dev/null
python tools/conv_probe.py 500 > $O/${tag}_conv_probe_f64.txt 2>/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_issue.hip -o tools/micro/mfma_f32_issue 2>/dev/null   # built here, never committed
./tools/micro/mfma_f32_issue > $O/${tag}_micro_mfma_f32_issue.txt 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I svgp-vae_amd/csrc tools/micro/sweep32_probe.hip -o tools/micro/sweep32_probe 2>/dev/null
./tools/micro/sweep32_probe > $O/${tag}_micro_sweep32.txt 2>/dev/null
python tools/sprites_gemm_f32_probe.py se 2>/dev/null | grep -v "^{" > $O/${tag}_sprites_gemm_f32_se.txt
for w in cfg2 cfg3 sp800 cfg5; do f=$(find $O/${tag}_prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_${w}_kernel_stats.csv; done
# the raw traces / counter dumps are large: keep the summaries only
rm -rf $O/${tag}_prof_* $O/${tag}_pmc_FETCH_SIZE $O/${tag}_pmc_WRITE_SIZE $O/${tag}_pmc_sq
echo "${2:-unknown}" > $O/${tag}_commit.txt      # the box has no .git: pass $(git rev-parse --short HEAD) as the 2nd argument
ls -la $O | grep ${tag}_
