#!/bin/bash
# Round-4 measurement set (run ON the GPU box, one call): bash tools/r4_profile.sh        -> gpurun_out/r4final/
#   bench lines (default, eager, the other configs, configs[4]'s share with and without activation checkpointing, torchrun N=1), rocprofv3 kernel
#   stats of the cycle step alone and of the default command, per-layer table, data-parallel evidence at one rank (where the RCCL kernels land;
#   host cost per graph cut), PMC traffic of the paired halo GEMM launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json; echo
python bench.py --graph 0 --no-cpu-baseline --no-f32-leg > $O/bench_eager.json 2>> $O/bench_cfg.err
python bench.py --mode reference --batch 1 --height 256 --width 256 --no-f32-leg > $O/bench_cfg1_256x256_b1_reference.json 2>> $O/bench_cfg.err
python bench.py --batch 4 --height 256 --width 256 --no-cpu-baseline --no-f32-leg > $O/bench_cfg2_256x256_b4.json 2>> $O/bench_cfg.err
python bench.py --batch 2 --height 512 --width 1024 --no-cpu-baseline --no-f32-leg > $O/bench_cfg5shape_1024x512_b2.json 2>> $O/bench_cfg.err
python bench.py --batch 2 --height 512 --width 1024 --no-cpu-baseline --no-f32-leg --checkpoint-blocks 1 > $O/bench_cfg5shape_1024x512_b2_checkpointed.json 2>> $O/bench_cfg.err
python bench.py --batch 2 --height 512 --width 1024 --no-cpu-baseline --no-f32-leg --no-reference-leg --graph 0 > $O/bench_cfg5shape_1024x512_b2_eager.json 2>> $O/bench_cfg.err
python bench.py --batch 2 --height 512 --width 1024 --no-cpu-baseline --no-f32-leg --no-reference-leg --graph 0 --checkpoint-blocks 1 > $O/bench_cfg5shape_1024x512_b2_eager_checkpointed.json 2>> $O/bench_cfg.err
python bench.py --no-cpu-baseline --no-f32-leg --no-reference-leg --checkpoint-blocks 1 > $O/bench_default_checkpointed.json 2>> $O/bench_cfg.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
echo "bench lines done"
python tools/layer_table.py > $O/layer_table.txt 2>&1
python tools/bench_in.py > $O/bench_in.txt 2>&1
python tools/graph_cut_cost.py > $O/graph_cut_cost.txt 2>&1
tail -4 $O/graph_cut_cost.txt
KSTAT_ROWS=80 bash tools/kstat.sh r4cycle bench.py --no-cpu-baseline --no-f32-leg --no-reference-leg > $O/cycle_step_only_kernel_shares.txt 2>&1
head -12 $O/cycle_step_only_kernel_shares.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-f32-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
rm -rf $O/stats
# data parallel at one rank under the profiler: the program itself after "--" (no launcher hop), rendezvous through the environment
(cd /tmp && export TMPDIR=/tmp && RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace --output-format csv -d $O/dptrace -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-reference-leg --no-f32-leg --no-kernel-timing > $O/bench_dp_n1_under_rocprof.json 2> $O/bench_dp_n1_under_rocprof.err)
python tools/dp_overlap_trace.py $(ls $O/dptrace/*/*kernel_trace.csv | head -1) > $O/dp_overlap_trace.txt 2>&1
rm -rf $O/dptrace
tail -30 $O/dp_overlap_trace.txt
# PMC: HBM traffic of the launches the paired cycle step makes
P=$O/pmcraw
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for op in fwd_pair dgrad_pair wgrad_pair2; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=$P/${op}_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/bench_conv.py --n 16 --ops $op --iters 8 > $d.log 2>&1 || echo "pass $op $c failed"
  done
done
python3 $R/tools/pmc_traffic.py $P fwd_pair,dgrad_pair,wgrad_pair2 16 > $O/traffic.json
rm -rf $P
cat $O/traffic.json | head -40
ls -la $O
