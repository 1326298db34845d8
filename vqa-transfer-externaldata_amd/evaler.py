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

    def eval(self):
        log.infov("Training starts")
        vocab = self.model.vocab
        answer_dict = self.model.answer_dict
        result_dict = {"qid2result": {}}
        avg_eval_report = {key: [] for key in self.model.report.keys()}
        avg_eval_report["testonly_score"] = []
        avg_eval_report["test_attr_only_score"] = []
        avg_eval_report["test_obj_only_score"] = []
        heavy_outputs = {key: [] for key in self.model.heavy_output.keys()}
        heavy_output_idx = 0
        if self.max_iter < 0:
            self.max_iter = 50000
        batch = self._first
        for s in range(self.max_iter):
            if batch is None:
                log.warning("OutOfRangeError happens at {} iter".format(s + 1))
                break
            self.model.set_batch(batch)      # (re)build: the constructor ran before the checkpoint was loaded
            self.model.build()
            torch.cuda.synchronize(self.model.device)
            reports = self.model.map_report(self.model.engine.report())
            outputs = {k: v.detach().cpu().numpy() for k, v in self.model.output.items()
                       if k not in ("att_score", "logit")}
            inputs = batch
            heavy_output = {k: v.detach().cpu().numpy() for k, v in self.model.heavy_output.items()} \
                if self.dump_heavy_output else None

            batch_size = len(inputs["id"])
            for b in range(batch_size):
                q_intseq = inputs["q_intseq"][b]
                q_intseq_len = inputs["q_intseq_len"][b]
                question = " ".join([vocab["vocab"][v] for v in q_intseq[:q_intseq_len]])
                id = int(inputs["id"][b])
                image_id = inputs["image_id"][b]
                pred = answer_dict["vocab"][int(outputs["pred"][b])]
                score = float(outputs["all_score"][b])
                max_train_score = float(outputs["max_train_score"][b])
                test_obj_score = float(outputs["test_obj_score"][b])
                test_obj_max_score = float(outputs["test_obj_max_score"][b])
                test_attr_score = float(outputs["test_attr_score"][b])
                test_attr_max_score = float(outputs["test_attr_max_score"][b])
                result_dict["qid2result"][id] = {
                    "image_id": image_id, "pred": pred, "question": question, "score": score,
                    "max_train_score": max_train_score, "test_obj_score": test_obj_score,
                    "test_obj_max_score": test_obj_max_score, "test_attr_score": test_attr_score,
                    "test_attr_max_score": test_attr_max_score,
                }
                if self.dump_heavy_output:
                    result_dict["qid2result"][id]["heavy_output_idx"] = heavy_output_idx
                    for key in heavy_output:
                        heavy_outputs[key].append(heavy_output[key][b])
                    heavy_output_idx += 1
                if max_train_score <= 0:
                    avg_eval_report["testonly_score"].append(score)
                    if test_obj_max_score <= 0:
                        avg_eval_report["test_attr_only_score"].append(test_attr_score)
                    if test_attr_max_score <= 0:
                        avg_eval_report["test_obj_only_score"].append(test_obj_score)
                for key in reports:          # appended once per SAMPLE, as the reference does (:167-168)
                    avg_eval_report[key].append(reports[key])
            batch = next(self._iter, None)

        result_dict["avg_eval_report"] = {
            key: np.array(avg_eval_report[key], dtype=np.float32).mean() for key in avg_eval_report}
        for key in avg_eval_report:
            result_dict["avg_eval_report"]["{}_num_point".format(key)] = len(avg_eval_report[key])
        log.info("saving pickle file to: {}".format(self.save_pkl))
        with open(self.save_pkl, "wb") as f:
            pickle.dump(result_dict, f)
        if self.dump_heavy_output:
            from . import hdf5_io
            hdf5_io.write(self.save_hdf5, {k: np.stack(v, axis=0) for k, v in heavy_outputs.items()})
        log.info("evaluation is done")
        return result_dict


def check_config(config):
    pass


def parse_checkpoint(config):
    config.ckpt_name = config.checkpoint.split("/")[-1]
    dirname = config.checkpoint.split("/")[-2]
    config.model_type = dirname.split("vqa_")[1].split("_d_")[0]
    qa_split_name = dirname.split("_d_")[1].split("_tf_record_memft")[0]
    config.tf_record_dir = os.path.join("data/preprocessed/vqa_v2", qa_split_name, "tf_record_memft")
    if "vfeat_bottomup_36_my" in dirname:
        config.vfeat_name = "vfeat_bottomup_36_my.hdf5"
    else:
        config.vfeat_name = "vfeat_bottomup_36.hdf5"
    config.vocab_path = os.path.join(config.tf_record_dir, config.vocab_name)
    config.vfeat_path = os.path.join(config.tf_record_dir, config.vfeat_name)


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--image_dir", type=str, default="data/VQA_v2/images", help=" ")
    parser.add_argument("--vocab_name", type=str, default="vocab.pkl", help=" ")
    parser.add_argument("--max_iter", type=int, default=-1, help=" ")
    parser.add_argument("--split", type=str, default="testval", help=" ",
                        choices=["train", "val", "testval", "test"])
    parser.add_argument("--prefix", type=str, default="default", help=" ")
    parser.add_argument("--checkpoint", type=str, default=None, required=True)
    parser.add_argument("--batch_size", type=int, default=512, help=" ")
    parser.add_argument("--debug", type=int, default=0, help="0: normal, 1: debug")
    parser.add_argument("--dump_heavy_output", action="store_true", default=False, help=" ")
    return parser


def main(argv=None):
    config = build_parser().parse_args(argv)
    check_config(config)
    parse_checkpoint(config)
    evaler = Evaler(config)
    evaler.eval()


if __name__ == "__main__":
    main()
