# Round-3 measurement pass, run on the GPU box:  gpurun -- bash profiles/collect_round.sh
# Full GPU tests, the default bench line (what the driver runs), rocprofv3 kernel stats of the same command (two-stream and
# serial), PMC passes (FETCH_SIZE / WRITE_SIZE; SQ counters of the GEMM kernels for both split arithmetics), the two baselines,
# the launch-bound B = 32 regime (eager and hipGraph).  Everything lands under gpurun_out/r3_*; profiles/summarise_round.py turns
# it into the committed files.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=r3
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -1 $O/${T}_smoke.log
python -m pytest tests -q -m gpu > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log; cp $O/parity.json $O/${T}_parity.json
cd $R && python bench.py --count-launches > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -c 300 $O/${T}_bench.json
python bench.py --shape bm --batch 32 --steps 300 --warmup 30 --alt-steps 0 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0 --count-launches > $O/${T}_bm32_eager.json 2> $O/${T}_bm32_eager.err
python bench.py --shape bm --batch 32 --steps 300 --warmup 30 --alt-steps 0 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0 --graph > $O/${T}_bm32_graph.json 2> $O/${T}_bm32_graph.err
python bench.py --precision bf16 --alt-steps 0 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0 > $O/${T}_bench_bf16.json 2> $O/${T}_bench_bf16.err
python bench.py --groups 6x10 --alt-steps 0 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0 > $O/${T}_bench_6x10.json 2> $O/${T}_bench_6x10.err
IGN_CONV_MATH=bf16x6 python bench.py --alt-steps 0 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0 > $O/${T}_bench_convx6.json 2> $O/${T}_bench_convx6.err
IGN_GEMM_MATH=bf16x6 python bench.py --config transformer --steps 6 --warmup 2 --cpu-sample 0 > $O/${T}_tr_x6.json 2> $O/${T}_tr_x6.err
for d in ResNet PatchTST TimesNet; do python bench.py --dnn $d --steps 8 --warmup 3 --alt-steps 0 --iso-steps 1 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0 > $O/${T}_ign_$d.json 2> $O/${T}_ign_$d.err; done
python bench.py --config transformer --precision bf16 --steps 6 --warmup 2 --cpu-sample 0 > $O/${T}_tr_bf16.json 2> $O/${T}_tr_bf16.err
cd /tmp && export TMPDIR=/tmp
P="--alt-steps 0 --baseline-steps 0 --cpu-sample 0 --harness-epochs 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_ign -o ign -- python $R/bench.py $P > $O/${T}_prof_ign.json 2> $O/${T}_prof_ign.err
export IGN_EXPERT_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_ign_serial -o ign -- python $R/bench.py --steps 8 --warmup 3 --iso-steps 1 $P > $O/${T}_prof_ign_serial.json 2> $O/${T}_prof_ign_serial.err
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_ign_$c -o ign -- python $R/bench.py --steps 4 --warmup 2 --iso-steps 1 $P > $O/${T}_pmc_ign_$c.json 2> $O/${T}_pmc_ign_$c.err
done
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for m in f16x3 bf16x6; do
IGN_CONV_MATH=$m rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_pmc_sq_$m -o ign -- python $R/bench.py --steps 4 --warmup 2 --iso-steps 1 $P > $O/${T}_pmc_sq_$m.json 2> $O/${T}_pmc_sq_$m.err
done
unset IGN_EXPERT_STREAMS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_eegcnn -o eegcnn -- python $R/bench.py --config eegcnn --steps 10 --warmup 3 --cpu-sample 0 > $O/${T}_prof_eegcnn.json 2> $O/${T}_prof_eegcnn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_tr -o tr -- python $R/bench.py --config transformer --steps 8 --warmup 3 --cpu-sample 0 > $O/${T}_prof_tr.json 2> $O/${T}_prof_tr.err
for m in f16x3 bf16x6; do
IGN_GEMM_MATH=$m rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_pmc_sq_tr_$m -o tr -- python $R/bench.py --config transformer --steps 2 --warmup 1 --cpu-sample 0 > $O/${T}_pmc_sq_tr_$m.json 2> $O/${T}_pmc_sq_tr_$m.err
done
ls $O | grep ${T}_ | head -60
