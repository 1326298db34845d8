"""Command line of the bridge that feeds `vlmap_answer_noc` / `_nocarch`: the counterpart of
vlmap_memft/export_noc_word_weights.py:1-95.

    python -m vqa_transfer_externaldata_amd.export_noc_word_weights --checkpoint train_dir/<run>/model-4801 [--data_dir DIR]

reads the two "no composition" heads `classifier_v/fc/{weights,biases}` and `classifier_l/fc/{weights,biases}` (:41-55) and the
three embedding tables from a checkpoint and writes `<checkpoint dir>/word_weights_<checkpoint name>/{weights.hdf5, vocab.pkl,
answer_dict.pkl}` with the datasets `v_word`, `l_word`, `l_answer_word`, `v_class_weights`, `v_class_biases`,
`l_class_weights`, `l_class_biases` (:72-80) -- what `modules.WordWeightAnswer(weight_name='v_class_weights', ...)` of
vqa/model_vlmap_answer_noc.py:190-202 looks up by answer string.  The checkpoint is a name -> tensor archive with the
reference's TF variable names (what this package's trainers save); the pre-training variant that TRAINS those two heads
(vlmap_memft/model_vlmap_noc_bf_or_wordset_withatt_sp.py) is a reference ablation outside this repo's scope, so such a
checkpoint comes from the reference side (converted) or from a test."""
from __future__ import annotations

import argparse
import os
import pickle

import numpy as np
import torch

from . import hdf5_io
from .export_word_weights import _load_pickle
from .log import log

DATASETS = (("v_word", "V_GloVe/embed_map"), ("l_word", "L_GloVe/embed_map"), ("l_answer_word", "LearnAnswerGloVe/embed_map"),
            ("v_class_weights", "classifier_v/fc/weights"), ("v_class_biases", "classifier_v/fc/biases"),
            ("l_class_weights", "classifier_l/fc/weights"), ("l_class_biases", "classifier_l/fc/biases"))


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
    missing = [name for _, name in DATASETS if name not in sd]
    if missing:
        raise KeyError("checkpoint has no %s (a `noc` pre-training checkpoint carries classifier_v / classifier_l)" % ", ".join(missing))
    A = len(answer_dict["vocab"])
    for scope in ("classifier_v", "classifier_l"):
        w = sd[scope + "/fc/weights"]
        if tuple(w.shape) != (config.class_feat_dim, A):
            raise ValueError("%s/fc/weights of the checkpoint is %s, expected [%d, %d] (--class_feat_dim x answers)"
                             % (scope, tuple(w.shape), config.class_feat_dim, A))
    log.warning("create directory: {}".format(config.save_dir))
    os.makedirs(config.save_dir)
    g = lambda k: np.asarray(sd[k].cpu() if torch.is_tensor(sd[k]) else sd[k])
    hdf5_io.write(os.path.join(config.save_dir, "weights.hdf5"), {ds: g(name) for ds, name in DATASETS})
    with open(os.path.join(config.save_dir, "vocab.pkl"), "wb") as f:
        pickle.dump(vocab, f)
    with open(os.path.join(config.save_dir, "answer_dict.pkl"), "wb") as f:
        pickle.dump(answer_dict, f)
    log.warning("weights are saved in: {}".format(os.path.join(config.save_dir, "weights.hdf5")))
    log.warning("done")
    return config.save_dir


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
