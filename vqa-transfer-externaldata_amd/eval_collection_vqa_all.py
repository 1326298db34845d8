"""Counterpart of vqa/eval_collection_vqa_all.py:1-115 -- eval_collection for the "VQA with seen answers in test" split:
every evaluated checkpoint's predictions are re-scored against `test_qid2anno.pkl` separately for each subset of
`test_detail_split.pkl` (split key -> list of question ids; :36-41, 65-83), giving `new_<key>_total_score`,
`new_<key>_obj_only_score`, `new_<key>_attr_only_score` beside the seven columns of eval_collection.

    python -m vqa_transfer_externaldata_amd.eval_collection_vqa_all --root_train_dir train_dir [--split test] [--qa_split_dir DIR]
"""
from __future__ import annotations

import argparse
import glob
import os
import pickle
from collections import defaultdict

import numpy as np

from .log import log


def _load(path_or_obj):
    if not isinstance(path_or_obj, str):
        return path_or_obj
    with open(path_or_obj, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:                        # python-2 cPickle file of the reference
            f.seek(0)
            return pickle.load(f, encoding="latin1")


def rescore_detail(qid2result, test_qid2anno, test_detail_split):
    """:65-83: per subset, the mean VQA score of the predictions over all its questions, over those whose ground truth has
    no test ATTRIBUTE answer and over those with no test OBJECT answer (np.mean of an empty list is nan, as there)"""
    mean = lambda v: float(np.array(v, dtype=np.float64).mean()) if len(v) else float("nan")
    score = lambda q: test_qid2anno[q]["answer_score"].get(qid2result[q]["pred"], 0)
    out = {}
    for key, qids in test_detail_split.items():
        out["new_{}_total_score".format(key)] = mean([score(q) for q in qids])
        out["new_{}_obj_only_score".format(key)] = mean([score(q) for q in qids if qid2result[q]["test_attr_max_score"] <= 0])
        out["new_{}_attr_only_score".format(key)] = mean([score(q) for q in qids if qid2result[q]["test_obj_max_score"] <= 0])
    return out


def collect(train_dir, split, test_qid2anno, test_detail_split):
    test_qid2anno, test_detail_split = _load(test_qid2anno), _load(test_detail_split)
    eval_dirs = glob.glob(os.path.join(train_dir, "model-*_eval_{}_*".format(split)))
    eval_iter2dir = {int(e.split("model-")[1].split("_eval")[0]): e for e in eval_dirs}
    collect_results = defaultdict(list)
    rows = [("iter", "testonly_score", "testonly_score_num_point", "test_obj_only_score", "test_obj_only_score_num_point",
             "test_attr_only_score", "test_attr_only_score_num_point")]
    for i in sorted(eval_iter2dir):
        with open(os.path.join(eval_iter2dir[i], "results.pkl"), "rb") as f:
            results = pickle.load(f)
        avg = results["avg_eval_report"]
        collect_results["iter"].append(i)
        for k, v in rescore_detail(results["qid2result"], test_qid2anno, test_detail_split).items():
            collect_results[k].append(v)
        row = ["{:05d}".format(i)]
        for key in ("testonly_score", "test_obj_only_score", "test_attr_only_score"):
            collect_results[key].append(avg[key])
            collect_results[key + "_num_point"].append(avg[key + "_num_point"])
            row += ["{:.5f}".format(avg[key]), "{:08d}".format(avg[key + "_num_point"])]
        rows.append(tuple(row))
    txt = os.path.join(train_dir, "collect_eval_{}_result.txt".format(split))
    with open(txt, "w") as f:
        for row in rows:
            f.write(" ".join(row) + "\n")
    with open(os.path.join(train_dir, "collect_eval_{}_result.pkl".format(split)), "wb") as f:
        pickle.dump(dict(collect_results), f)
    log.warning("result is saved in {}".format(txt))
    return dict(collect_results)


def build_parser():
    """flags of vqa/eval_collection_vqa_all.py:12-22"""
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--root_train_dir", type=str, default=None, help=" ")
    parser.add_argument("--train_dirs", nargs="+", type=str, default=[], help=" ")
    parser.add_argument("--split", type=str, default="test", help=" ", choices=["train", "val", "testval", "test"])
    parser.add_argument("--qa_split_dir", type=str, default="data/preprocessed/vqa_v2"
                        "/qa_split_objattr_answer_3div4_genome_memft_check_all_answer_thres1_50000_thres2_-1_with_seen_answer_in_test",
                        help=" ")
    return parser


def main(argv=None):
    config = build_parser().parse_args(argv)
    if config.root_train_dir is None and len(config.train_dirs) == 0:
        raise ValueError("Set either root_train_dir or train_dirs")
    if config.root_train_dir is not None and len(config.train_dirs) > 0:
        raise ValueError("Do not set both root_train_dir and train_dirs")
    dirs = config.train_dirs if config.root_train_dir is None else glob.glob(os.path.join(config.root_train_dir, "vqa_*"))
    log.warning("loading target data ..")
    anno = _load(os.path.join(config.qa_split_dir, "test_qid2anno.pkl"))
    detail = _load(os.path.join(config.qa_split_dir, "test_detail_split.pkl"))
    log.warning("loading target data is done")
    out = {}
    for i, d in enumerate(sorted(dirs)):
        log.warning("[{:02d}] train_dir: {}".format(i, d))
        out[d] = collect(d, config.split, anno, detail)
    return out


if __name__ == "__main__":
    main()
