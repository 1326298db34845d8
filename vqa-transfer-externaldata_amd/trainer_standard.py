"""Counterpart of vqa/trainer_standard.py:1-310 -- "Trainer for standard train/val split data".

The same loop as vqa/trainer.py on the standard VQA split, with the differences of that file: only `standard` (and, through
its own `get_model_class`, `standard_word2vec`) models (:17-24, 299-300); run directory `./train_dir/std_<model>_<data>_
<prefix>_<hyper>_<time>` (no seed, :38-40); train and val splits only (:47-60, 102-104); gradients clipped at global norm
**0.25** instead of 20 (:95); no transfer loading, no word-weight directory (:122, 154); the loop runs for 1 000 000 steps
(:158, 165) unless `--max_train_iter` (an addition here) says otherwise; default data `bottomup_vqa_tf_record_memft` /
`vfeat_bottomup_36.hdf5` (:282-284)."""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import trainer as _T

CLIP_GRADIENTS = 0.25          # vqa/trainer_standard.py:95
MAX_STEPS = 1000000            # vqa/trainer_standard.py:158


class Trainer(_T.Trainer):

    @staticmethod
    def get_model_class(model_type="standard"):
        if model_type == "standard":
            from .model_standard import Model
        elif model_type == "standard_word2vec":
            from .model_standard_word2vec import Model
        else:
            raise ValueError("Unknown model_type")
        return Model

    def __init__(self, config, datasets=None, image_features=None):
        if getattr(config, "train_dir", None) is None:
            dataset_str = "d_" + "_".join(config.tf_record_dir.replace("data/preprocessed/vqa_v2/", "").split("/"))
            dataset_str += "_" + config.vfeat_name.replace(".hdf5", "")
            hyper = "bs{}_lr{}".format(config.batch_size, config.learning_rate)
            config.train_dir = "./train_dir/std_{}_{}_{}_{}_{}".format(config.model_type, dataset_str, config.prefix, hyper,
                                                                       time.strftime("%Y%m%d-%H%M%S"))
        for k, v in (("ft_vlmap", False), ("vlmap_word_weight_dir", None), ("pretrained_param_path", None), ("seed", 123),
                     ("debug", 0), ("max_train_iter", MAX_STEPS)):
            if not hasattr(config, k):
                setattr(config, k, v)
        if datasets is not None:                       # train / val only (:47-60)
            datasets = {k: v for k, v in datasets.items() if k in ("train", "val")}
        super().__init__(config, datasets=datasets, image_features=image_features)
        for split in ("testval", "test"):
            self._iters.pop(split, None)
        self.model.engine.clip_norm = CLIP_GRADIENTS


def build_parser():
    """flags of vqa/trainer_standard.py:275-302 (+ --max_train_iter, --sort_by_length, --device_batch_cache_gb)"""
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--image_dir", type=str, default="data/VQA_v2/images", help=" ")
    parser.add_argument("--tf_record_dir", type=str, default="data/preprocessed/vqa_v2/bottomup_vqa_tf_record_memft", help=" ")
    parser.add_argument("--vfeat_name", type=str, default="vfeat_bottomup_36.hdf5", help=" ")
    parser.add_argument("--vocab_name", type=str, default="vocab.pkl", help=" ")
    parser.add_argument("--train_average_iter", type=int, default=200)
    parser.add_argument("--val_average_iter", type=int, default=419)
    parser.add_argument("--heavy_summary_step", type=int, default=800)
    parser.add_argument("--validation_step", type=int, default=800)
    parser.add_argument("--checkpoint_step", type=int, default=800)
    parser.add_argument("--prefix", type=str, default="default", help=" ")
    parser.add_argument("--checkpoint", type=str, default=None)
    parser.add_argument("--learning_rate", type=float, default=0.001, help=" ")
    parser.add_argument("--lr_weight_decay", action="store_true", default=False)
    parser.add_argument("--batch_size", type=int, default=512, help=" ")
    parser.add_argument("--model_type", type=str, default="standard", help=" ", choices=["standard"])
    parser.add_argument("--max_train_iter", type=int, default=MAX_STEPS, help="(not in the reference: its loop is 1000000 steps)")
    parser.add_argument("--sort_by_length", type=int, default=1, help="(not in the reference; results are unchanged)")
    parser.add_argument("--device_batch_cache_gb", type=int, default=32, help="(not in the reference)")
    return parser


def parse_config(argv=None):
    config = build_parser().parse_args(argv)
    config.vocab_path = os.path.join(config.tf_record_dir, config.vocab_name)
    config.vfeat_path = os.path.join(config.tf_record_dir, config.vfeat_name)
    return config


def main(argv=None):
    config = parse_config(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
        config.device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    torch.manual_seed(123)
    np.random.seed(123)
    Trainer(config).train()


if __name__ == "__main__":
    main()
