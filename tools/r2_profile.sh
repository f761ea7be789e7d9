#!/bin/bash
# Round-2 measurement set (run ON the GPU box, one call): bash tools/r2_profile.sh
#   gpurun_out/r2final/: bench lines (default, eager, reference mode, configs 1/2/5 shapes, torchrun N=1, mixed), rocprofv3
#   kernel stats of the default bench command, the per-layer table, PMC passes (HBM traffic, LDS conflicts / MFMA busy, norms)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2final
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --graph 0 --no-cpu-baseline > $O/bench_eager.json 2>> $O/bench_cfg.err
python bench.py --mixed 1 --no-cpu-baseline > $O/bench_mixed.json 2>> $O/bench_cfg.err
python bench.py --mode reference --batch 1 --height 256 --width 256 > $O/bench_cfg1_256x256_b1_reference.json 2>> $O/bench_cfg.err
python bench.py --batch 4 --height 256 --width 256 --no-cpu-baseline > $O/bench_cfg2_256x256_b4.json 2>> $O/bench_cfg.err
python bench.py --batch 2 --height 512 --width 1024 --no-cpu-baseline > $O/bench_cfg5shape_1024x512_b2.json 2>> $O/bench_cfg.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
python tools/layer_table.py > $O/layer_table.txt 2>&1
python tools/enqueue_time.py --graph 0 > $O/enqueue.txt 2>&1
python tools/enqueue_time.py --graph 1 >> $O/enqueue.txt 2>&1
python tools/bench_in.py > $O/bench_in.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
rm -rf $O/stats
bash $R/tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp $R/gpurun_out/pmc/traffic.json $O/traffic.json
bash $R/tools/pmc_lds.sh > $O/pmc_lds.txt 2>&1
bash $R/tools/pmc_in.sh > $O/pmc_in.log 2>&1
cp $R/gpurun_out/pmc_in/in_traffic.json $O/in_traffic.json
rm -rf $R/gpurun_out/pmc $R/gpurun_out/pmc_lds $R/gpurun_out/pmc_in
ls -la $O
