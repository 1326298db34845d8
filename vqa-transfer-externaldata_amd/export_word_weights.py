"""Command line of the pre-training -> VQA bridge: the counterpart of vlmap_memft/export_word_weights.py:1-82.

    python -m vqa_transfer_externaldata_amd.export_word_weights --checkpoint train_dir/<run>/model-4801 [--data_dir DIR]

writes `<checkpoint dir>/word_weights_<checkpoint name>/{weights.hdf5, vocab.pkl, answer_dict.pkl}` (refusing to
overwrite, like the reference :20-27); `--vlmap_word_weight_dir` of the VQA trainer points at that directory.
`--class_feat_dim` is accepted for flag compatibility and checked against the checkpoint's classifier."""
from __future__ import annotations

import argparse
import os
import pickle

import torch

from .log import log
from .pretrain import export_word_weights


def _load_pickle(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:          # pickles written by the reference's python 2
            f.seek(0)
            return pickle.load(f, encoding="latin1")


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--data_dir", type=str,
                        default="data/preprocessed/visualgenome/memft_all_new_vocab50_obj3000_attr1000_maxlen10", help=" ")
    parser.add_argument("--class_feat_dim", type=int, default=2048, help=" ")
    parser.add_argument("--checkpoint", type=str, required=True, help="ex) ./model-1")
    return parser


def run(config, vocab=None, answer_dict=None):
    ckpt_dir, ckpt_name = os.path.dirname(config.checkpoint), os.path.basename(config.checkpoint)
    config.save_dir = os.path.join(ckpt_dir, "word_weights_{}".format(ckpt_name))
    if os.path.exists(config.save_dir):
        raise ValueError("Do not overwrite: {}".format(config.save_dir))
    vocab = vocab if vocab is not None else _load_pickle(os.path.join(config.data_dir, "vocab.pkl"))
    answer_dict = answer_dict if answer_dict is not None else _load_pickle(os.path.join(config.data_dir, "answer_dict.pkl"))
    log.info("Checkpoint path: {}".format(config.checkpoint))
    sd = torch.load(config.checkpoint, map_location="cpu")
    w = sd["classifier/fc/weights"]
    if tuple(w.shape) != (config.class_feat_dim, len(answer_dict["vocab"])):
        raise ValueError("classifier/fc/weights of the checkpoint is %s, expected [%d, %d] (--class_feat_dim x answers)"
                         % (tuple(w.shape), config.class_feat_dim, len(answer_dict["vocab"])))
    log.warning("create directory: {}".format(config.save_dir))
    d = export_word_weights(sd, vocab, answer_dict, config.save_dir)
    log.warning("weights are saved in: {}".format(os.path.join(d, "weights.hdf5")))
    log.warning("done")
    return d


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
