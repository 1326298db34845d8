"""MI355X counterpart of vqa/evaler.py: same flags (:214-232), checkpoint-path grammar
(parse_checkpoint, :195-211), eval-dir naming (:66-76), result schema of results.pkl
(:147-179) and the is_train=False / dropout-still-on behaviour (F6)."""
from __future__ import annotations

import argparse
import os
import pickle
import time

import numpy as np
import torch

from . import importer, input_ops_vqa
from .log import log


class Evaler(object):

    @staticmethod
    def get_model_class(model_type="vlmap_answer"):
        return importer.get_model_class(model_type)

    def __init__(self, config, set_checkpoint=True, image_features=None, data=None):
        self.config = config
        self.split = config.split
        self.max_iter = config.max_iter
        self.dump_heavy_output = config.dump_heavy_output
        self.vfeat_path = config.vfeat_path
        self.tf_record_dir = config.tf_record_dir

        self.batch_size = config.batch_size
        self._iter = input_ops_vqa.create(self.batch_size, self.tf_record_dir, self.split, is_train=False,
                                          scope="{}_ops".format(self.split), shuffle=False, data=data)
        self._first = next(self._iter, None)
        if self._first is None:
            raise ValueError("split %s is empty" % self.split)

        Model = self.get_model_class(config.model_type)
        log.infov("using model class: {}".format(Model))
        config.global_batch = None
        self.model = Model(self._first, config, is_train=False, image_features=image_features)
        if set_checkpoint:
            self.set_eval_dir(config)
            self.load_checkpoint(config)

    def set_eval_dir(self, config):
        self.checkpoint = config.checkpoint
        self.eval_dir = config.checkpoint + "_eval_{}".format(self.split)
        if self.dump_heavy_output:
            self.eval_dir += "_dump_heavy"
        self.eval_dir += "_{}".format(time.strftime("%Y%m%d-%H%M%S"))
        if not os.path.exists(self.eval_dir):
            os.makedirs(self.eval_dir)
        log.infov("Eval Dir: %s", self.eval_dir)
        self.save_hdf5 = os.path.join(self.eval_dir, "results.hdf5")  # heavy outputs (vqa/evaler.py:75,181-186)
        self.save_pkl = os.path.join(self.eval_dir, "results.pkl")

    def load_checkpoint(self, config):
        if config.checkpoint is not None:
            log.info("Checkpoint path: {}".format(config.checkpoint))
            sd = torch.load(config.checkpoint, map_location="cpu")
            self.model.engine.load_state_dict(sd)
            log.info("Loaded the checkpoint")
        log.warning("Evaluation initialization is done")

    # ------------------------------------------------------------------ evaluation pass
    # Per-sample outputs of the model that go into results.pkl (vqa/evaler.py:129-160), result key -> model.output key
    _RESULT_FIELDS = (("score", "all_score"), ("max_train_score", "max_train_score"),
                      ("test_obj_score", "test_obj_score"), ("test_obj_max_score", "test_obj_max_score"),
                      ("test_attr_score", "test_attr_score"), ("test_attr_max_score", "test_attr_max_score"))

    def _launch(self, batch, slot):
        """Queue the forward pass of `batch` and the device->host copies of everything the result file needs into the
        pinned staging buffers of `slot`; returns a handle for _collect.  Nothing here waits for the GPU."""
        model = self.model
        model.set_batch(batch)               # (re)build: the constructor ran before the checkpoint was loaded
        model.build()
        n = len(batch["id"])
        stage = self._stage[slot]
        want = {k: model.output[src] for k, src in self._RESULT_FIELDS}
        want["pred"] = model.output["pred"]
        want["report"] = model.engine.tensor("report")[:13]
        if self.dump_heavy_output:
            want.update({"heavy/" + k: v for k, v in model.heavy_output.items()})
        host = {}
        for k, t in want.items():
            buf = stage.get(k)
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = stage[k] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            buf.copy_(t, non_blocking=True)  # stream-ordered after the forward, before the next batch overwrites the workspace
            host[k] = buf
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(model.device))
        return batch, n, host, done

    def _collect(self, handle, acc):
        """Host side of one batch, run while the GPU works on the next one: whole-batch array operations, no per-sample
        Python arithmetic.  Fills acc['qid2result'] and the running statistics of avg_eval_report."""
        batch, n, host, done = handle
        done.synchronize()
        col = {k: host[k].numpy().astype(np.float64) for k, _ in self._RESULT_FIELDS}
        pred_words = self._answers[host["pred"].numpy().astype(np.int64)]
        tokens = self._words[np.asarray(batch["q_intseq"], dtype=np.int64)]            # [n, T] vocabulary strings
        lens = np.asarray(batch["q_intseq_len"], dtype=np.int64)
        questions = [" ".join(row[:m]) for row, m in zip(tokens.tolist(), lens.tolist())]
        qids = np.asarray(batch["id"]).astype(np.int64).tolist()
        names = ("image_id", "pred", "question") + tuple(k for k, _ in self._RESULT_FIELDS)
        columns = [list(batch["image_id"]), pred_words.tolist(), questions] + [col[k].tolist() for k, _ in self._RESULT_FIELDS]
        records = [dict(zip(names, row)) for row in zip(*columns)]
        if self.dump_heavy_output:
            base = acc["n_heavy"]
            for i, rec in enumerate(records):
                rec["heavy_output_idx"] = base + i
            for k in acc["heavy"]:
                acc["heavy"][k].append(host["heavy/" + k].numpy().copy())
            acc["n_heavy"] = base + n
        acc["qid2result"].update(zip(qids, records))
        # test-only statistics (:162-166): questions none of whose answers is a training answer, and among them the
        # ones that only have attribute / object test answers
        unseen = col["max_train_score"] <= 0
        acc["extra"]["testonly_score"].append(col["score"][unseen])
        acc["extra"]["test_attr_only_score"].append(col["test_attr_score"][unseen & (col["test_obj_max_score"] <= 0)])
        acc["extra"]["test_obj_only_score"].append(col["test_obj_score"][unseen & (col["test_attr_max_score"] <= 0)])
        # the batch's report scalars count once per SAMPLE, as in the reference (:167-168)
        rep = host["report"].numpy()
        scalars = self.model.map_report({self._report_keys[i]: float(rep[i]) for i in range(13)})
        for k, v in scalars.items():
            acc["report"].setdefault(k, []).append((v, n))

    def eval(self):
        log.infov("Evaluation of split {} starts".format(self.split))
        model = self.model
        self._words = np.asarray(model.vocab["vocab"], dtype=object)
        self._answers = np.asarray(model.answer_dict["vocab"], dtype=object)
        self._report_keys = [model.engine.lib.vqa_report_key(i).decode() for i in range(13)]
        self._stage = ({}, {})
        acc = {"qid2result": {}, "report": {k: [] for k in model.report},
               "extra": {"testonly_score": [], "test_attr_only_score": [], "test_obj_only_score": []},
               "heavy": {k: [] for k in model.heavy_output}, "n_heavy": 0}
        limit = self.max_iter if self.max_iter >= 0 else 50000
        batch, in_flight, it = self._first, None, 0
        t_start = time.time()
        while True:
            handle = None
            if batch is not None and it < limit:
                handle = self._launch(batch, it & 1)       # the GPU starts on batch `it` ...
                it += 1
            if in_flight is not None:
                self._collect(in_flight, acc)              # ... while the host writes up batch `it - 1`
            if handle is None:
                if batch is None and it < limit:
                    log.warning("OutOfRangeError happens at {} iter".format(it + 1))
                break
            in_flight = handle
            batch = next(self._iter, None)
        elapsed = time.time() - t_start

        series = {k: (np.repeat(np.array([v for v, _ in pairs], np.float32), [m for _, m in pairs])
                      if pairs else np.zeros([0], np.float32)) for k, pairs in acc["report"].items()}
        for k, chunks in acc["extra"].items():
            series[k] = np.concatenate(chunks).astype(np.float32) if chunks else np.zeros([0], np.float32)
        avg = {k: v.mean() for k, v in series.items()}           # (an empty selection gives nan, as np.mean of [] does)
        avg.update({"{}_num_point".format(k): len(v) for k, v in series.items()})
        result_dict = {"qid2result": acc["qid2result"], "avg_eval_report": avg}
        self.eval_seconds, self.eval_samples = elapsed, len(acc["qid2result"])
        log.info("saving pickle file to: {}".format(self.save_pkl))
        with open(self.save_pkl, "wb") as f:
            pickle.dump(result_dict, f)
        if self.dump_heavy_output:
            from . import hdf5_io
            hdf5_io.write(self.save_hdf5, {k: np.concatenate(v, axis=0) for k, v in acc["heavy"].items()})
        log.info("evaluation is done: {} questions in {:.2f} s".format(self.eval_samples, elapsed))
        return result_dict


_SPLIT_ROOT = "data/preprocessed/vqa_v2"


def parse_checkpoint(config):
    """Everything but the checkpoint file is encoded in the name of its run directory (vqa/evaler.py:195-211),
    `vqa_<model_type>_d_<qa split>_tf_record_memft_<vfeat name>_...`: model type, tf-record directory, feature file."""
    run_dir, config.ckpt_name = config.checkpoint.split("/")[-2:]
    after_prefix = run_dir.split("vqa_")[1]
    config.model_type, rest = after_prefix.split("_d_")[0], run_dir.split("_d_")[1]
    config.tf_record_dir = os.path.join(_SPLIT_ROOT, rest.split("_tf_record_memft")[0], "tf_record_memft")
    config.vfeat_name = "vfeat_bottomup_36%s.hdf5" % ("_my" if "vfeat_bottomup_36_my" in run_dir else "")
    config.vocab_path = os.path.join(config.tf_record_dir, config.vocab_name)
    config.vfeat_path = os.path.join(config.tf_record_dir, config.vfeat_name)


# flags and defaults of vqa/evaler.py:214-232 (the contract of the command line)
_FLAGS = (("--image_dir", dict(type=str, default="data/VQA_v2/images")),
          ("--vocab_name", dict(type=str, default="vocab.pkl")),
          ("--max_iter", dict(type=int, default=-1)),
          ("--split", dict(type=str, default="testval", choices=["train", "val", "testval", "test"])),
          ("--prefix", dict(type=str, default="default")),
          ("--checkpoint", dict(type=str, default=None, required=True)),
          ("--batch_size", dict(type=int, default=512)),
          ("--debug", dict(type=int, default=0, help="0: normal, 1: debug")),
          ("--dump_heavy_output", dict(action="store_true", default=False)))


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **dict({"help": " "}, **kw))
    return parser


def main(argv=None):
    config = build_parser().parse_args(argv)
    parse_checkpoint(config)
    Evaler(config).eval()


if __name__ == "__main__":
    main()
