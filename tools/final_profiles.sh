#!/bin/bash
# Round-end measurement session on the GPU box: everything the committed summaries under profiles/ are made from.
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh'
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fusion --no-secondary"
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
python3 $R/tools/sweep_readout.py --out $O/sweep_readout.txt --csv $O/readout_kernels.csv > $O/sweep_readout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/readout_stats -- python3 $R/tools/sweep_readout.py --out $O/sweep_readout_prof.txt > $O/readout_stats.log 2>&1
python3 $R/tools/sweep_kq.py --out $O/sweep_kq.txt > $O/sweep_kq.log 2>&1
python3 $R/tools/sweep_default.py > $O/sweep_default.txt 2>&1
python3 $R/tools/probe_tile_12.py 28 --pairs > $O/tile12.txt 2>&1
python3 $R/tools/probe_fused_blocks.py 5 > $O/fused_blocks_k5.txt 2>&1
python3 $R/tools/probe_rdm.py > $O/rdm.txt 2>&1
# keep only the small summaries of the profiler directories (the traces are hundreds of MiB)
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
echo "all done"
