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


def rescore(qid2result, pure_test_qid2anno):
    """vqa/eval_collection.py:59-70: the prediction strings of one evaluation re-scored against the FULL annotation
    of the test-only questions (`pure_test_qid2anno.pkl`: qid -> {'answer_score': {answer string: VQA score}}):
    mean score over all of them, over those whose ground truth has no test ATTRIBUTE answer (object-only) and over
    those with no test OBJECT answer (attribute-only).  A question missing from the results raises KeyError, as in
    the reference; an empty subset gives nan, as np.mean([]) does there."""
    import numpy as np

    def mean(vals):
        return float(np.array(vals, dtype=np.float64).mean()) if len(vals) else float("nan")

    score = lambda qid, anno: anno["answer_score"].get(qid2result[qid]["pred"], 0)
    return {"new_testonly_score": mean([score(q, a) for q, a in pure_test_qid2anno.items()]),
            "new_test_obj_only_score": mean([score(q, a) for q, a in pure_test_qid2anno.items()
                                             if qid2result[q]["test_attr_max_score"] <= 0]),
            "new_test_attr_only_score": mean([score(q, a) for q, a in pure_test_qid2anno.items()
                                              if qid2result[q]["test_obj_max_score"] <= 0])}


def collect(train_dir, split="testval", pure_test_qid2anno=None):
    """vqa/eval_collection.py:41-105 for one run directory: per evaluated iteration the testonly / obj-only /
    attr-only scores (+ numbers of points) into collect_eval_<split>_result.txt (the reference's seven columns) and
    .pkl; with `pure_test_qid2anno` (the dict, or the path of pure_test_qid2anno.pkl) the pickle also carries the
    re-scored new_testonly_score / new_test_obj_only_score / new_test_attr_only_score (:59-76)."""
    if isinstance(pure_test_qid2anno, str):
        with open(pure_test_qid2anno, "rb") as f:
            try:
                pure_test_qid2anno = pickle.load(f)
            except UnicodeDecodeError:                        # python-2 cPickle file of the reference
                f.seek(0)
                pure_test_qid2anno = pickle.load(f, encoding="latin1")
    eval_dirs = glob.glob(os.path.join(train_dir, "model-*_eval_{}_*".format(split)))
    eval_iter2dir = {int(e.split("model-")[1].split("_eval")[0]): e for e in eval_dirs}
    collect_results = defaultdict(list)
    collect_list = [("iter", "testonly_score", "testonly_score_num_point", "test_obj_only_score",
                     "test_obj_only_score_num_point", "test_attr_only_score", "test_attr_only_score_num_point")]
    for i in sorted(eval_iter2dir):
        with open(os.path.join(eval_iter2dir[i], "results.pkl"), "rb") as f:
            results = pickle.load(f)
        avg = results["avg_eval_report"]
        collect_results["iter"].append(i)
        if pure_test_qid2anno is not None:
            for k, v in rescore(results["qid2result"], pure_test_qid2anno).items():
                collect_results[k].append(v)
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


def build_parser():
    """flags of vqa/eval_collection.py:12-21"""
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--root_train_dir", type=str, default=None, help=" ")
    parser.add_argument("--train_dirs", nargs="+", type=str, default=[], help=" ")
    parser.add_argument("--split", type=str, default="test", help=" ", choices=["train", "val", "testval", "test"])
    parser.add_argument("--qa_split_dir", type=str, default="data/preprocessed/vqa_v2"
                        "/qa_split_objattr_answer_3div4_genome_memft_check_all_answer_thres1_50000_thres2_-1", help=" ")
    return parser


def main(argv=None):
    config = build_parser().parse_args(argv)
    if config.root_train_dir is None and len(config.train_dirs) == 0:
        raise ValueError("Set either root_train_dir or train_dirs")
    if config.root_train_dir is not None and len(config.train_dirs) > 0:
        raise ValueError("Do not set both root_train_dir and train_dirs")
    dirs = config.train_dirs if config.root_train_dir is None else glob.glob(os.path.join(config.root_train_dir, "vqa_*"))
    anno = os.path.join(config.qa_split_dir, "pure_test_qid2anno.pkl")
    out = {}
    for i, d in enumerate(sorted(dirs)):
        log.warning("[{:02d}] train_dir: {}".format(i, d))
        out[d] = collect(d, config.split, pure_test_qid2anno=anno)
    return out


if __name__ == "__main__":
    main()
