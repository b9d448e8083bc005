#!/usr/bin/env python3
"""CLI driver -- drop-in for IGN/run.py: every flag of IGN/run.py:17-135 with its default and meaning (including
the inverted ``--amp``: passing it turns bf16 autocast OFF), the seed loop ``[0,42,1234,8237,2023]``
(:564), checkpoint skip/load (:580-602), test + pickled results (:605-625).

Differences, all repairs of fork defects listed in SURVEY section 0:
  * ``--data`` defaults to ``UEA`` and ``--data_root`` to ``./data/UEA_multivariate`` (the upstream defaults that
    survive as comments in IGN/run.py:68-69), so ``run_uea.sh`` -- which passes neither -- works (D2);
  * ``--data SYNTH`` (+ ``--synthetic n,C,T,classes``) is the synthetic CHISCO-shaped provider of the benchmark;
  * multi-GPU is one process per GPU: ``python -m torch.distributed.run --nproc-per-node N run.py ...``.
"""
import argparse
import os
import pickle
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np
import torch

from exp.experiment_classification import Experiment as ClassificationExperiment

exp_dict = {
    "classification": ClassificationExperiment,
    "regression": None,      # Monash regression twin: out of scope; key kept for the registry
}


def build_parser():
    p = argparse.ArgumentParser()
    # data
    p.add_argument("--data", type=str, default="UEA", choices=['EEG', 'EEG3', 'UEA', 'SYNTH'])
    p.add_argument("--data_root", type=str, default="./data/UEA_multivariate")
    p.add_argument("--json_path", type=str, default="./json/textmaps.json")
    p.add_argument("--hipgraph", action="store_true",
                   help="replay the training step (forward, fused loss, backward, Adam) as one captured hipGraph -- for the "
                        "launch-bound regime of small batches (run_uea.sh: --batch_size 32): one host call per step instead of "
                        "~45 launches.  Ignored (eager) with gradient accumulation / clipping, bf16 autocast or several ranks")
    p.add_argument("--synthetic", type=str, default=None, help="SYNTH provider: n,C,T,classes (default 8192,122,1000,3)")
    p.add_argument("--target_channels", type=int, default=122)
    p.add_argument("--target_timepoints", type=int, default=1651)
    p.add_argument("--max_files", type=int, default=1000)
    p.add_argument("--max_subjects", type=int, default=5)
    p.add_argument("--subject_id", type=str, default="sub-01")
    p.add_argument("--subject_ids", type=str, nargs='+', default=["sub-01,sub-02,sub-03"])
    p.add_argument("--task_type", type=str, default="imagine", choices=['imagine', 'read', 'both'])
    # EEG-CNN baseline
    p.add_argument("--eegcnn_layers", type=int, default=2)
    p.add_argument("--eegcnn_pooling", type=str, default='mean', choices=[None, 'mean', 'sum', 'top'])
    p.add_argument("--eegcnn_cnn_f1", type=int, default=8)
    p.add_argument("--eegcnn_cnn_f2", type=int, default=8)
    p.add_argument("--eegcnn_kernel1", type=int, default=125)
    p.add_argument("--eegcnn_kernel2", type=int, default=25)
    p.add_argument("--eegcnn_pool1", type=int, default=2)
    p.add_argument("--eegcnn_pool2", type=int, default=5)
    p.add_argument("--eegcnn_dropout1", type=float, default=0.1)
    p.add_argument("--eegcnn_dropout2", type=float, default=0.1)
    p.add_argument("--eegcnn_n_heads", type=int, default=8)
    p.add_argument("--eegcnn_d_ff", type=int, default=256)
    # SBM / InterpGN
    p.add_argument("--model", type=str, default='InterpGN', choices=['SBM', 'LTS', 'InterpGN', 'DNN', 'EEGCNN'])
    p.add_argument("--dnn_type", type=str, default='Transformer',
                   choices=['FCN', 'Transformer', 'TimesNet', 'PatchTST', 'ResNet'])
    p.add_argument("--dataset", type=str, default="BasicMotions")
    p.add_argument("--lambda_reg", type=float, default=0.1)
    p.add_argument("--lambda_div", type=float, default=0.1)
    p.add_argument("--epsilon", type=float, default=1.)
    p.add_argument("--num_shapelet", type=int, default=10)
    p.add_argument("--gating_value", type=float, default=None)
    p.add_argument("--pos_weight", action="store_true")
    p.add_argument("--sbm_cls", type=str, default='linear')
    p.add_argument("--distance_func", type=str, default='euclidean')
    p.add_argument("--beta_schedule", type=str, default='constant')
    p.add_argument("--memory_efficient", action="store_true")
    # experiment
    p.add_argument("--lr", type=float, default=5e-3)
    p.add_argument("--lr_decay", action="store_true")
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--gradient_clip", type=float, default=0)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument('--log_interval', type=int, default=20)
    p.add_argument("--min_epochs", type=int, default=0)
    p.add_argument("--train_epochs", type=int, default=500)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--patience", type=int, default=50)
    p.add_argument("--multi_gpu", action='store_true')
    p.add_argument("--test_only", action='store_true')
    p.add_argument("--seed", type=int, default=-1)
    p.add_argument("--amp", action='store_false', default=True)     # sic: the flag turns autocast OFF (D3)
    # basic config
    p.add_argument('--task_name', type=str, default='classification')
    p.add_argument('--model_id', type=str, default='test')
    p.add_argument('--embed', type=str, default='timeF')
    p.add_argument('--freq', type=str, default='h')
    # DNN configs
    p.add_argument('--top_k', type=int, default=5)
    p.add_argument('--num_kernels', type=int, default=6)
    p.add_argument('--enc_in', type=int, default=7)
    p.add_argument('--dec_in', type=int, default=7)
    p.add_argument('--c_out', type=int, default=7)
    p.add_argument('--d_model', type=int, default=512)
    p.add_argument('--n_heads', type=int, default=8)
    p.add_argument('--e_layers', type=int, default=2)
    p.add_argument('--d_layers', type=int, default=1)
    p.add_argument('--d_ff', type=int, default=2048)
    p.add_argument('--moving_avg', type=int, default=25)
    p.add_argument('--factor', type=int, default=1)
    p.add_argument('--distil', action='store_false', default=True)
    p.add_argument('--dropout', type=float, default=0)
    p.add_argument('--activation', type=str, default='gelu')
    p.add_argument('--output_attention', action='store_true')
    p.add_argument('--label_len', type=int, default=48)
    p.add_argument('--pred_len', type=int, default=96)
    p.add_argument('--seasonal_patterns', type=str, default='Monthly')
    p.add_argument('--inverse', action='store_true', default=False)
    return p


def get_args(argv=None):
    args = build_parser().parse_args(argv)
    args.root_path = args.data_root if args.data in ('EEG', 'EEG3', 'SYNTH') else f"{args.data_root}/{args.dataset}"
    args.is_training = True
    return args


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def init_distributed():
    """One process per GPU when launched by torch.distributed.run; RCCL ('nccl') over xGMI."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    return dist.get_rank()


def main(argv=None):
    args = get_args(argv)
    exp_cls = exp_dict[args.task_name]
    if exp_cls is None:
        raise SystemExit(f"task_name={args.task_name!r} is not part of this build (classification only)")
    rank = init_distributed()
    if args.data in ('EEG', 'EEG3', 'UEA') and not os.path.exists(args.root_path):
        raise SystemExit(f"data path does not exist: {args.root_path}")
    seeds = [0, 42, 1234, 8237, 2023] if args.seed == -1 else [args.seed]
    for i, seed in enumerate(seeds):
        set_seed(seed)
        args.seed = seed
        print(f"===== experiment {i + 1}/{len(seeds)} - seed {seed} =====")
        experiment = exp_cls(args=args)
        if rank == 0:
            experiment.print_args()
        ckpt = f"{experiment.checkpoint_dir}/checkpoint.pth"
        if not args.test_only:
            if os.path.exists(ckpt):
                print(f"checkpoint exists, skipping training: {ckpt}")
            else:
                experiment.train()
                torch.cuda.empty_cache()
        elif not os.path.exists(ckpt):
            print(f"warning: checkpoint missing, nothing to test: {ckpt}")
            continue
        if os.path.exists(ckpt):
            experiment.model.load_state_dict(torch.load(ckpt, map_location=experiment.device, weights_only=True))
        else:
            print("warning: testing a randomly initialised model")
        test_loss, test_metrics, test_df = experiment.test(save_csv=True, result_dir=f"./result/{args.model}")
        if rank == 0 and test_metrics is not None:
            with open(f"{os.path.dirname(ckpt)}/test_results.pkl", 'wb') as f:
                pickle.dump({'test_loss': test_loss, 'test_metrics': test_metrics, 'test_df': test_df,
                             'args': vars(args)}, f)
            print(f"accuracy: {test_metrics.accuracy:.4f}  loss: {test_metrics.loss:.4f}")


if __name__ == "__main__":
    main()
