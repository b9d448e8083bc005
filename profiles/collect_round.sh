# Round measurement pass run on the GPU box (gpurun -- bash profiles/collect_round.sh): full GPU tests, default bench line,
# rocprofv3 kernel stats of the same command (two-stream and serial), PMC FETCH_SIZE / WRITE_SIZE passes, baseline profiles.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python -m pytest tests -q -m gpu > $O/r2k_tests.log 2>&1; tail -3 $O/r2k_tests.log; cp $O/parity.json $O/r2k_parity.json
cd $R && python bench.py > $O/r2k_bench.json 2> $O/r2k_bench.err; tail -c 200 $O/r2k_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2k_prof_ign -o ign -- python $R/bench.py --baseline-steps 0 --cpu-sample 0 > $O/r2k_prof_ign.json 2> $O/r2k_prof_ign.err
export IGN_EXPERT_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2k_prof_ign_serial -o ign -- python $R/bench.py --steps 8 --warmup 3 --alt-steps 0 --iso-steps 1 --baseline-steps 0 --cpu-sample 0 > $O/r2k_prof_ign_serial.json 2> $O/r2k_prof_ign_serial.err
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r2k_pmc_ign_$c -o ign -- python $R/bench.py --steps 4 --warmup 2 --alt-steps 0 --iso-steps 1 --baseline-steps 0 --cpu-sample 0 > $O/r2k_pmc_ign_$c.json 2> $O/r2k_pmc_ign_$c.err
done
unset IGN_EXPERT_STREAMS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2k_prof_eegcnn -o eegcnn -- python $R/bench.py --config eegcnn --steps 10 --warmup 3 --cpu-sample 0 > $O/r2k_prof_eegcnn.json 2> $O/r2k_prof_eegcnn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2k_prof_tr -o tr -- python $R/bench.py --config transformer --steps 8 --warmup 3 --cpu-sample 0 > $O/r2k_prof_tr.json 2> $O/r2k_prof_tr.err
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r2k_pmc_eegcnn_$c -o eegcnn -- python $R/bench.py --config eegcnn --steps 4 --warmup 2 --cpu-sample 0 > $O/r2k_pmc_eegcnn_$c.json 2> $O/r2k_pmc_eegcnn_$c.err
done
ls $O | grep r2k | head -40
