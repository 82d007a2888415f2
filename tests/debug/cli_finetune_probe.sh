#!/bin/bash
# scratch: the two train.py runs of tests/test_gpu_cli.py::test_train_then_detect, N times over, printing the logged losses
# (is the fine-tune step after three chaotic steps at BATCH_SIZE 4 near overflow?)   usage: cli_finetune_probe.sh [runs] [bn decay] [learning rate]
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
for i in $(seq 1 ${1:-3}); do
  d=/tmp/cliprobe_$i; rm -rf $d; mkdir -p $d
  cat > $d/config.yaml <<EOC
NUM_BBOXES_PER_CELL : 5
MAX_NUM_BBOXES : 13
LOCATION_LOSS_ALPHA : 1000.0
BATCH_SIZE : 4
INPUT_SIZE : 299
NUM_TRAIN_EXAMPLES : 56945
NUM_TRAIN_ITERATIONS : 1000000
LOG_EVERY_N_STEPS : 1
${2:+BATCHNORM_MOVING_AVERAGE_DECAY : $2}
${3:+INITIAL_LEARNING_RATE : $3}
EOC
  python -c "from multibox_amd import priors as PR; PR.save_priors('$d/priors.pkl', PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]))"
  python train.py --priors $d/priors.pkl --logdir $d/log --config $d/config.yaml --max_number_of_steps 3 --synthetic > $d/o1.txt 2>&1
  python train.py --priors $d/priors.pkl --logdir $d/log --config $d/config.yaml --max_number_of_steps 4 --synthetic --fine_tune > $d/o2.txt 2>&1
  echo "run $i rc=$?"; cat $d/log/train_log.jsonl | cut -c1-220; grep -i "error" $d/o2.txt | tail -3
done
