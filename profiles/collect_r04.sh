#!/bin/bash
# Round-4 profiles.  Run on the GPU box from the repo root:  bash profiles/collect_r04.sh
# (build/libmbpe_diag.so must exist: tools/mkvar.sh diag -DMBPE_DIAG; build/lds_atomic_floor: see tools/lds_atomic_floor.hip)
# Writes the condensed summaries to gpurun_out/r04/ (copied into profiles/ afterwards).
# rocprofv3: counters in their own runs (never together with traces), program directly after "--".
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$R/profiles/summarize.py

# 1. the driver's command: kernel trace + stats (a step is a whole training: 25 trainings)
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_under_profiler.json 2> $O/kt.err || { tail -5 $O/kt.err; exit 1; }
python3 $S stats $(find $O/kt -name '*kernel_stats.csv' | head -1) > $O/r04_kernel_stats.csv
python3 $S trace $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/r04_kernel_trace.csv
python3 $S pc $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/r04_pair_count_launches.csv
rm -rf $O/kt
echo "step 1 done"; date

# 2. config 3 (bible stand-in)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -o kb -- python3 $R/bench.py --config bible --steps 20 --warmup 5 > $O/r04_bible_bench_under_profiler.json 2> $O/kb.err || { tail -5 $O/kb.err; exit 1; }
python3 $S trace $(find $O/kb -name '*kernel_trace.csv' | head -1) > $O/r04_bible_kernel_trace.csv
rm -rf $O/kb
echo "step 2 done"; date

# 3. HBM traffic, one counter per run (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2): three whole trainings
export MBPE_TRAFFIC_COMMAND="bench.py --steps 2 --warmup 1 --no-full-run --no-cpu-baseline (three whole trainings: every fused pass of a run, 4.29e9 -> 2.8e9 slots)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --pmc $c --output-format csv -d $O/p_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-full-run --no-cpu-baseline > $O/p_$c.log 2>&1 || { tail -5 $O/p_$c.log; exit 1; }
  python3 $S pmc $(find $O/p_$c -name '*counter_collection.csv' | head -1) > $O/r04_pmc_$c.csv
  rm -rf $O/p_$c
done
python3 $R/profiles/make_traffic_json.py $O r04 > $O/r04_pmc_traffic.json
echo "step 3 done"; date

# 4. SQ counters (instruction mix and LDS bank conflicts of the pair-count scan and the fused pass)
i=0
: > $O/r04_pmc_sq.csv
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s_$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-full-run --no-cpu-baseline > $O/s_$i.log 2>&1 || { tail -5 $O/s_$i.log; continue; }
  python3 $S sq $(find $O/s_$i -name '*counter_collection.csv' | head -1) >> $O/r04_pmc_sq.csv
  if [ $i = 1 ]; then cp $(find $O/s_1 -name '*counter_collection.csv' | head -1) $O/set1_counters.csv; cp $(find $O/s_1 -name '*kernel_trace.csv' | head -1) $O/set1_trace.csv; fi
  rm -rf $O/s_$i
done
echo "step 4 done"; date

# 5. the floors of a fused pass: timing-only instantiations of the kernel (-DMBPE_DIAG) on the same passes
cd $R
MBPE_LIB=$R/build/libmbpe_diag.so timeout -k 10 300 python3 tools/fused_diag.py > $O/r04_fused_diag.log 2>&1 || tail -5 $O/r04_fused_diag.log
python3 profiles/make_floor_json.py --diag-log $O/r04_fused_diag.log --counters $O/set1_counters.csv --trace $O/set1_trace.csv > $O/r04_fused_floor.json
rm -f $O/set1_counters.csv $O/set1_trace.csv
cat $O/r04_fused_floor.json
echo "step 5 done"; date

# 6. pair-count scan through the C-ABI (random and text), what the LDS alone allows it, a training sequence by
#    sequence, and the bytes a multi-GPU run would exchange
timeout -k 10 300 python3 tools/pc_time.py > $O/r04_pc_time.log 2>&1
timeout -k 10 120 build/lds_atomic_floor > $O/r04_lds_atomic_floor.log 2>&1
timeout -k 10 300 python3 tools/seq_sizes.py > $O/r04_seq_sizes.json 2> $O/ss.err || tail -3 $O/ss.err
timeout -k 10 300 python3 tools/exchange_bytes.py > $O/r04_exchange_bytes.json 2> $O/xb.err || tail -3 $O/xb.err
ls -la $O
