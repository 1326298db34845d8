"""Evaluate every checkpoint of one or more training runs: the counterpart of vqa/eval_multiple_model.py:40-130.

    python -m vqa_transfer_externaldata_amd.eval_multiple_model --root_train_dir train_dir [--split test ...]
    python -m vqa_transfer_externaldata_amd.eval_multiple_model --train_dirs train_dir/vqa_..._A train_dir/vqa_..._B

Same flags and defaults as the reference (:42-60).  Runs without checkpoints are skipped; model type, tf_record
directory and feature-file name are parsed out of the first checkpoint's directory name (evaler.parse_checkpoint);
the feature table is read ONCE and shared by every Evaler, and a run whose name points at another feature file is
skipped with a warning (:101-120).  Checkpoints here are single `model-<iter>` files (torch.save), not TensorFlow's
`model-<iter>.index` triples."""
from __future__ import annotations

import argparse
import glob
import os

from . import eval_collection, evaler
from .log import log
from .model_vlmap_answer import load_image_features


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--image_dir", type=str, default="data/VQA_v2/images", help=" ")
    parser.add_argument("--vocab_name", type=str, default="vocab.pkl", help=" ")
    parser.add_argument("--max_iter", type=int, default=-1, help=" ")
    parser.add_argument("--split", type=str, default="test", help=" ", choices=["train", "val", "testval", "test"])
    parser.add_argument("--prefix", type=str, default="default", help=" ")
    parser.add_argument("--root_train_dir", type=str, default=None, help=" ")
    parser.add_argument("--train_dirs", nargs="+", type=str, default=[], help=" ")
    parser.add_argument("--batch_size", type=int, default=512, help=" ")
    parser.add_argument("--dump_heavy_output", action="store_true", default=False, help=" ")
    parser.add_argument("--debug", type=int, default=0, help="0: normal, 1: debug")
    return parser


def train_dirs_of(config):
    """:62-86 -- `vqa_*` under --root_train_dir (or --train_dirs as given), sorted, runs without checkpoints dropped"""
    if config.root_train_dir is None:
        dirs = list(config.train_dirs)
    else:
        dirs = glob.glob(os.path.join(config.root_train_dir, "vqa_*"))
    dirs = sorted(dirs)
    log.warning("all_train_dirs:")
    for i, d in enumerate(dirs):
        log.infov("{:02d}: {}".format(i, d))
    return [d for d in dirs if eval_collection.checkpoints_of(d)]


def features_dict(path):
    feats, spat, boxes, nb, max_box, dim = load_image_features(path)
    return {"features": feats, "spatials": spat, "normal_boxes": boxes, "num_boxes": nb, "max_box_num": max_box,
            "vfeat_dim": dim}


def run(config, image_features=None, data=None):
    """image_features / data: optional in-memory stand-ins for the feature file and the tfrecord splits (tests)"""
    dirs = train_dirs_of(config)
    if not dirs:
        raise ValueError("no train_dir with checkpoints (model-<iter>) under the given directories")
    config.checkpoint = eval_collection.checkpoints_of(dirs[-1])[0]        # the reference initialises from the LAST
    evaler.parse_checkpoint(config)                                        # directory it scanned (:76-89)
    loaded_vfeat_path = config.vfeat_path
    if image_features is None:
        log.infov("loading image features...")
        image_features = features_dict(config.vfeat_path)
        log.infov("done")
    done = {}
    for train_dir in dirs:
        ckpts = eval_collection.checkpoints_of(train_dir)
        config.checkpoint = ckpts[0]
        evaler.parse_checkpoint(config)
        if loaded_vfeat_path != config.vfeat_path:
            log.warning("vfeat_path for this train_dir is different from the initialized one: {} vs {}".format(
                loaded_vfeat_path, config.vfeat_path))
            continue
        for i, ckpt in enumerate(ckpts):
            log.warning("evaluate {} [{}/{}]: {}".format(os.path.basename(ckpt), i, len(ckpts), ckpt))
            c = argparse.Namespace(**vars(config))
            c.checkpoint = ckpt
            evaler.parse_checkpoint(c)
            ev = evaler.Evaler(c, image_features=image_features, data=data)
            done[ckpt] = ev.eval()
    log.warning("all evaluation is done")
    return done


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
