#!/bin/bash
# Round-2 profiles.  Run on the GPU box from the repo root:  bash profiles/collect_r02.sh
# Writes the condensed summaries to gpurun_out/r02/ (copied into profiles/ afterwards).
# rocprofv3: counters in their own runs (never together with traces), program directly after "--".
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$R/profiles/summarize.py

# 1. the driver's command: kernel trace + stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_bench_under_profiler.json 2> $O/kt.err || { tail -5 $O/kt.err; exit 1; }
python3 $S stats $(find $O/kt -name '*kernel_stats.csv' | head -1) > $O/r02_kernel_stats.csv
python3 $S trace $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/r02_kernel_trace.csv
python3 $S trace20 $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/r02_kernel_trace_timed_region.csv
rm -rf $O/kt

# 2. config 3 (bible stand-in)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -o kb -- python3 $R/bench.py --config bible --steps 20 --warmup 5 > $O/r02_bible_bench_under_profiler.json 2> $O/kb.err || { tail -5 $O/kb.err; exit 1; }
python3 $S trace $(find $O/kb -name '*kernel_trace.csv' | head -1) > $O/r02_bible_kernel_trace.csv
rm -rf $O/kb

# 3. HBM traffic, one counter per run (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/p_$c -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-full-run --no-cpu-baseline > $O/p_$c.log 2>&1 || { tail -5 $O/p_$c.log; exit 1; }
  python3 $S pmc $(find $O/p_$c -name '*counter_collection.csv' | head -1) > $O/r02_pmc_$c.csv
  rm -rf $O/p_$c
done

# 4. SQ counters (LDS bank conflicts of the pair-count scan, instruction mix of the fused pass)
i=0
: > $O/r02_pmc_sq.csv
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $set --output-format csv -d $O/s_$i -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-full-run --no-cpu-baseline > $O/s_$i.log 2>&1 || { tail -5 $O/s_$i.log; continue; }
  python3 $S sq $(find $O/s_$i -name '*counter_collection.csv' | head -1) >> $O/r02_pmc_sq.csv
  rm -rf $O/s_$i
done
python3 $R/profiles/make_traffic_json.py $O > $O/r02_pmc_traffic.json
ls -la $O
