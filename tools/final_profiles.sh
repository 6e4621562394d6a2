#!/bin/bash
# Round-end measurement session on the GPU box: everything the committed summaries under profiles/ are made from.
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh [part]'      part = bench | sweeps | gkp | all (default)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
PART=${1:-all}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fusion --no-secondary"
if [ "$PART" = bench ] || [ "$PART" = all ]; then
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fusion --no-secondary > $O/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 $B > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 $B > $O/prof_write.log 2>&1
echo "bench profiles done"
python3 $R/bench.py --config cfg4 --steps 3 --warmup 1 > $O/cfg4.json 2> $O/cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_stats -- python3 $R/bench.py --config cfg4 --steps 1 --warmup 1 > $O/cfg4_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cfg4_fetch -- python3 $R/bench.py --config cfg4 --steps 1 --warmup 0 > $O/cfg4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cfg4_write -- python3 $R/bench.py --config cfg4 --steps 1 --warmup 0 > $O/cfg4_write.log 2>&1
echo "cfg4 profiles done"
python3 $R/bench.py --config cfg5 --qubits-per-gpu 30 --steps 1 --warmup 1 > $O/cfg5_n30.json 2> $O/cfg5_n30.err
# the N > 1 code path on this one GPU (ranks share GPU 0, send / recv staged through the host: timings meaningless)
cd $R
QSV_BENCH_ONE_GPU=1 python3 bench.py --gpus 4 --scaling strong --steps 2 --warmup 1 > $O/rehearsal_strong4.json 2> $O/rehearsal_strong4.err
QSV_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --config cfg5 --qubits-per-gpu 24 --steps 2 --warmup 1 > $O/rehearsal_cfg5.json 2> $O/rehearsal_cfg5.err
QSV_BENCH_ONE_GPU=1 python3 bench.py --gpus 4 --config cfg3 --qubits-per-gpu 26 --steps 2 --warmup 1 > $O/rehearsal_cfg3.json 2> $O/rehearsal_cfg3.err
cd /tmp
echo "rehearsals done"
fi
if [ "$PART" = sweeps ] || [ "$PART" = all ]; then
python3 $R/tools/sweep_readout.py --out $O/sweep_readout.txt --csv $O/readout_kernels.csv > $O/sweep_readout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/readout_stats -- python3 $R/tools/sweep_readout.py --out $O/sweep_readout_prof.txt > $O/readout_stats.log 2>&1
python3 $R/tools/sweep_kq.py --out $O/sweep_kq.txt > $O/sweep_kq.log 2>&1
python3 $R/tools/sweep_default.py > $O/sweep_default.txt 2>&1
python3 $R/tools/probe_complex_product.py 28 > $O/complex_product.txt 2>&1
python3 $R/tools/probe_rdm.py 28 > $O/rdm.txt 2>&1
python3 $R/tools/bench_percall.py > $O/percall.txt 2>&1
for m in 0 1 2; do for r in 0 8; do QSV_COPY_MODE=$m QSV_COPY_REGIONS=$r python3 $R/tools/probe_copy.py 28 >> $O/copy.txt 2>&1; done; done
for n in 25 26 27 29 31; do python3 $R/tools/probe_tile_12.py $n > $O/tile12_n$n.txt 2>&1; done
python3 $R/tools/probe_tile_12.py 28 --pairs > $O/tile12_n28_pairs.txt 2>&1
python3 $R/tools/probe_fused_blocks.py 5 > $O/fused_blocks_k5.txt 2>&1
python3 $R/tools/bench_gkp.py --circuit grover27 --bond 100 --rel-err 1e-2 --out $O/gkp_grover.json > $O/gkp_grover.log 2>&1
echo "sweeps done"
fi
if [ "$PART" = gkp ] || [ "$PART" = all ]; then
python3 $R/tools/bench_gkp.py --circuit grover27 --bond 100 --rel-err 1e-2 --out $O/gkp_grover.json > $O/gkp_grover.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gkp_stats -o gkp -- python3 $R/tools/bench_gkp.py --circuit grover27 --bond 100 --rel-err 1e-2 > $O/gkp_stats.log 2>&1
python3 $R/tools/probe_exact_split.py > $O/exact_split.txt 2>&1
python3 $R/tools/probe_sequence.py 30 3 > $O/sequence_blocks.txt 2>&1
echo "gkp done"
fi
# keep only the small summaries of the profiler directories (the traces are hundreds of MiB)
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
echo "all done"
