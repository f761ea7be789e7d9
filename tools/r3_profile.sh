#!/bin/bash
# Round-3 measurement set, part 1 (run ON the GPU box, one call): bash tools/r3_profile.sh
#   gpurun_out/r3final/: bench lines (default, eager, other configs, torchrun N=1), rocprofv3 kernel stats of the cycle step alone and of
#   the default command, the per-layer table, norm / class-map micro-benchmarks.   Part 2 (PMC passes): tools/r3_pmc.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --graph 0 --no-cpu-baseline --no-f32-leg > $O/bench_eager.json 2>> $O/bench_cfg.err
python bench.py --mode reference --batch 1 --height 256 --width 256 --no-f32-leg > $O/bench_cfg1_256x256_b1_reference.json 2>> $O/bench_cfg.err
python bench.py --batch 4 --height 256 --width 256 --no-cpu-baseline --no-f32-leg > $O/bench_cfg2_256x256_b4.json 2>> $O/bench_cfg.err
python bench.py --batch 2 --height 512 --width 1024 --no-cpu-baseline --no-f32-leg > $O/bench_cfg5shape_1024x512_b2.json 2>> $O/bench_cfg.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
python tools/layer_table.py > $O/layer_table.txt 2>&1
python tools/bench_in.py > $O/bench_in.txt 2>&1
python tools/bench_seg.py > $O/bench_seg.txt 2>&1
KSTAT_ROWS=80 bash tools/kstat.sh r3cycle bench.py --no-cpu-baseline --no-f32-leg --no-reference-leg > $O/cycle_step_only_kernel_shares.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-f32-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
rm -rf $O/stats
ls -la $O
