"""Per-checkpoint evaluation sweep and score collection: the counterparts of
vqa/eval_multiple_model.py:40-130 (evaluate every `model-<iter>` of a run with the feature table loaded
once) and vqa/eval_collection.py:41-105 (collect testonly / obj-only / attr-only scores per iteration
into `collect_eval_<split>_result.{txt,pkl}`)."""
from __future__ import annotations

import argparse
import glob
import os
import pickle
import re
from collections import defaultdict

from . import evaler
from .log import log


def checkpoints_of(train_dir):
    """`model-<iter>` files of a run, sorted by iteration."""
    out = {}
    for p in glob.glob(os.path.join(train_dir, "model-*")):
        m = re.fullmatch(r"model-(\d+)", os.path.basename(p))
        if m and os.path.isfile(p):
            out[int(m.group(1))] = p
    return [out[k] for k in sorted(out)]


def eval_all_checkpoints(config, train_dir, image_features=None, data=None):
    """vqa/eval_multiple_model.py: one Evaler per checkpoint; the feature table (image_features) is shared."""
    results = {}
    for ckpt in checkpoints_of(train_dir):
        c = argparse.Namespace(**vars(config))
        c.checkpoint = ckpt
        ev = evaler.Evaler(c, image_features=image_features, data=data)
        results[ckpt] = ev.eval()
        log.infov("evaluated %s -> %s", ckpt, ev.eval_dir)
    return results


def collect(train_dir, split="testval"):
    """vqa/eval_collection.py:41-105 for one run directory (without the qid2anno re-scoring, which needs
    the VQA annotation pickles)."""
    eval_dirs = glob.glob(os.path.join(train_dir, "model-*_eval_{}_*".format(split)))
    eval_iter2dir = {int(e.split("model-")[1].split("_eval")[0]): e for e in eval_dirs}
    collect_results = defaultdict(list)
    collect_list = [("iter", "testonly_score", "testonly_score_num_point", "test_obj_only_score",
                     "test_obj_only_score_num_point", "test_attr_only_score", "test_attr_only_score_num_point")]
    for i in sorted(eval_iter2dir):
        with open(os.path.join(eval_iter2dir[i], "results.pkl"), "rb") as f:
            avg = pickle.load(f)["avg_eval_report"]
        collect_results["iter"].append(i)
        row = ["{:05d}".format(i)]
        for key in ("testonly_score", "test_obj_only_score", "test_attr_only_score"):
            collect_results[key].append(avg[key])
            collect_results[key + "_num_point"].append(avg[key + "_num_point"])
            row += ["{:.5f}".format(avg[key]), "{:08d}".format(avg[key + "_num_point"])]
        collect_list.append(tuple(row))
    txt = os.path.join(train_dir, "collect_eval_{}_result.txt".format(split))
    with open(txt, "w") as f:
        for row in collect_list:
            f.write(" ".join(row) + "\n")
    with open(os.path.join(train_dir, "collect_eval_{}_result.pkl".format(split)), "wb") as f:
        pickle.dump(dict(collect_results), f)
    log.warning("result is saved in {}".format(txt))
    return dict(collect_results)
