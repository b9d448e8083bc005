#!/bin/bash
# Launcher of the UEA multivariate benchmark -- same loop and flags as IGN/run_uea.sh:52-70
# (30 datasets x 5 seeds; --amp = fp32, see run.py).  Run from this directory: `bash run_uea.sh [dataset ...]`.
MODEL="InterpGN"
DNN_TYPE="FCN"
NUM_SHAPELET=10
LAMBDA_DIV=0.1
LAMBDA_REG=0.1
EPS=1.
BETA_SCHEDULE="constant"
GATING_VALUE=1

UEA_DATASETS=(
    "ArticularyWordRecognition" "AtrialFibrillation" "BasicMotions" "CharacterTrajectories" "Cricket"
    "ERing" "Epilepsy" "EthanolConcentration" "FaceDetection" "FingerMovements" "HandMovementDirection"
    "Handwriting" "Heartbeat" "InsectWingbeat" "JapaneseVowels" "LSST" "Libras" "NATOPS" "PenDigits"
    "PhonemeSpectra" "RacketSports" "SelfRegulationSCP1" "SelfRegulationSCP2" "SpokenArabicDigits"
    "UWaveGestureLibrary" "StandWalkJump"
    # many variates: large shapelet banks
    "PEMS-SF" "DuckDuckGeese"
    # very long series (stride rule of Shapelet.py:162 applies from seq_len 3000)
    "MotorImagery" "EigenWorms"
)
if [ "$#" -gt 0 ]; then UEA_DATASETS=("$@"); fi

SEEDS=(${SEEDS:-0 42 1234 8237 2023})
EPOCHS=${EPOCHS:-500}
DATA_ROOT=${DATA_ROOT:-./data/UEA_multivariate}

for dataset in "${UEA_DATASETS[@]}"; do
    for seed in "${SEEDS[@]}"; do
        python run.py \
            --model $MODEL \
            --dnn_type $DNN_TYPE \
            --data_root $DATA_ROOT \
            --dataset $dataset \
            --train_epochs $EPOCHS \
            --batch_size 32 \
            --lr 5e-3 \
            --dropout 0. \
            --num_shapelet $NUM_SHAPELET \
            --lambda_div $LAMBDA_DIV \
            --lambda_reg $LAMBDA_REG \
            --epsilon $EPS \
            --beta_schedule $BETA_SCHEDULE \
            --seed $seed \
            --gating_value $GATING_VALUE \
            --amp
    done
done
