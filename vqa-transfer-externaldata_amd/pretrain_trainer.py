"""Counterpart of vlmap_memft/trainer.py (pre-training stage of BASELINE config 5): same flags and
defaults (:323-351), two splits (train / val), loop cadence (:202-263), log line, checkpoints
`model-<step>` every checkpoint_step; `export_word_weights` turns a checkpoint into the
`word_weights_model-N/` directory the VQA trainer's --vlmap_word_weight_dir expects (run.py:252-286)."""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch

from . import dataset_vlmap
from .log import log
from .model_vlmap_bf_or_wordset_withatt_sp import Model
from .pretrain import export_word_weights  # noqa: F401  (re-exported: the bridge lives with the engine)

MODEL_TYPES = ["vlmap_bf_or_wordset_withatt_sp"]


class Trainer(object):

    @staticmethod
    def get_model_class(model_type="vlmap_bf_or_wordset_withatt_sp"):
        if model_type not in MODEL_TYPES:
            raise ValueError("model_type %r is a pre-training ablation that is out of scope (supported: %s)"
                             % (model_type, ", ".join(MODEL_TYPES)))
        return Model

    def __init__(self, config, dataset):
        self.config = config
        # Data parallel (BASELINE configs[4]: bs 512 over 8 GPUs; the reference runs one process, vlmap_memft/trainer.py:
        # 129-137): every rank draws the SAME global batch (same seed), keeps its contiguous shard of the images, and
        # the gradients meet in a bucketed all-reduce overlapped with the backward phases (pretrain.PretrainEngine).
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        self._allreduce = None
        if self.world > 1:
            from . import dp
            self._allreduce = dp.BucketedAllReduce()
        hyper = "bs{}_lr{}".format(config.batch_size, config.learning_rate)
        self.train_dir = getattr(config, "train_dir", None) or "./train_dir/{}_{}_{}_{}".format(
            config.model_type, config.prefix, hyper, time.strftime("%Y%m%d-%H%M%S"))
        if self.rank == 0:
            os.makedirs(self.train_dir, exist_ok=True)
            log.infov("Train Dir: %s", self.train_dir)
        self.batch_size = config.batch_size
        # Input side (not in the reference, whose sequential py_func pipeline fed a 2018 GPU): the feature tables of
        # both splits stay in HBM and batches carry image indices (features_on_device), batches are assembled by
        # forked producer processes (input_workers) or one producer thread (input_prefetch) while the GPU runs the
        # step.  Producers are started here, before anything has initialised the GPU.
        resident = bool(getattr(config, "features_on_device", 1))
        workers = int(getattr(config, "input_workers", 0) or 0)
        prefetch = int(getattr(config, "input_prefetch", 2) or 0)
        mk = lambda split, shuffle, name: dataset_vlmap.create_ops(
            self.batch_size, dataset[split], is_train=True, scope=name, shuffle=shuffle, seed=config.seed,
            resident=resident, workers=workers, prefetch=prefetch)
        self._iters = {"train": mk("train", True, "train_ops"), "val": mk("val", False, "val_ops")}
        if resident:
            tr, va = dataset["train"], dataset["val"]
            n_train = int(np.asarray(tr.image_features).shape[0])
            same = va.image_features is tr.image_features
            cat = (lambda a, b: a if same else np.concatenate([np.asarray(a), np.asarray(b)], 0))
            config.feature_tables = (cat(tr.image_features, va.image_features),
                                     cat(tr.spatial_features, va.spatial_features),
                                     cat(np.asarray(tr.num_boxes), np.asarray(va.num_boxes)))
            if not same:       # the validation split's rows follow the training split's in the device tables
                def offset(it):
                    for b in it:
                        b["image_idx"] = b["image_idx"] + n_train
                        yield b
                self._iters["val"] = offset(self._iters["val"])
        self._pending = self._shard(next(self._iters["train"]))
        self.model = self.get_model_class(config.model_type)(self._pending, config, is_train=True)
        self.global_step = 0
        self.learning_rate = config.learning_rate
        self.max_train_iter = config.max_train_iter
        self.train_average_iter, self.val_average_iter = config.train_average_iter, config.val_average_iter
        self.heavy_summary_step, self.validation_step = config.heavy_summary_step, config.validation_step
        self.checkpoint_step = config.checkpoint_step
        self._summary_path = os.path.join(self.train_dir, "summaries.jsonl")
        if config.checkpoint is not None:
            # parameters + Adam slots + step count (tf.train.Saver restores the slots and beta powers too)
            sd = torch.load(config.checkpoint, map_location="cpu")
            self.model.engine.load_state_dict(sd)
            self.global_step = int(sd.get("global_step", 0))
            self.model._step = int(sd.get("dropout_step", self.global_step))

    def _lr(self):
        if self.config.lr_weight_decay:
            return self.learning_rate * (0.5 ** (self.global_step // 10000))
        return self.learning_rate

    def _shard(self, batch):
        """This rank's images of the global batch + what the step needs to know about the whole: the global row of its
        first image (dropout stream), the global image count, and the global valid-entry counts per category (the
        denominators of the masked mean losses), all from the host arrays -- no collective."""
        if self.world == 1:
            return batch
        from . import dp
        n_img = len(batch["image_idx"] if "image_idx" in batch else batch["image_ft"])
        lo, hi = dp.shard_bounds(n_img, self.rank, self.world)
        n_entries = int(self.config.data_cfg.n_obj_bf)
        gv = tuple(float(np.clip(np.asarray(batch[k + "_blank_fill/num"]), 0, n_entries).sum()) for k in ("obj", "attr"))
        out = {k: (v[lo:hi] if hasattr(v, "__len__") and not isinstance(v, (str, bytes, dict)) and len(v) == n_img else v)
               for k, v in batch.items()}
        out["_dp"] = {"row_offset": lo, "global_rows": n_img, "global_valid": gv}
        return out

    def _next(self, split):
        if split == "train" and self._pending is not None:
            b, self._pending = self._pending, None
            return b
        return self._shard(next(self._iters[split]))

    def run_train_step(self, use_heavy_summary):
        _start = time.time()
        prepared, self._prepared = getattr(self, "_prepared", None), None
        if prepared is None:
            prepared = self.model.prepare(self._next("train"))
        self.model.build(prepared=prepared, defer_report=True)
        self.model.backward(reducer=self._allreduce)
        self.model.apply_gradients(self._lr())
        # the GPU is busy with this step: draw the next batch and upload it now (tf.data prefetch of the reference)
        self._prepared = self.model.prepare(self._next("train"))
        self.model.finish_report()
        torch.cuda.synchronize(self.model.device)
        self.global_step += 1
        report = dict(self.model.report)
        summary = {"split": "train", "step": self.global_step, **report} if use_heavy_summary else None
        return self.global_step, summary, report["total_loss"], report, time.time() - _start

    def run_val_step(self, use_heavy_summary):
        _start = time.time()
        self.model.set_batch(self._next("val"))     # (a prepared train batch, if any, stays queued for the next train step)
        self.model.build()
        torch.cuda.synchronize(self.model.device)
        report = dict(self.model.report)
        summary = {"split": "val", "step": self.global_step, **report} if use_heavy_summary else None
        return self.global_step, summary, report["total_loss"], report, time.time() - _start

    def add_summary(self, s):
        if s is not None and self.rank == 0:
            with open(self._summary_path, "a") as f:
                f.write(json.dumps(s) + "\n")

    def save_checkpoint(self):
        path = os.path.join(self.train_dir, "model-{}".format(self.global_step))
        if self.rank != 0:          # every rank holds the same parameters and Adam slots
            return path
        sd = self.model.engine.state_dict()
        sd["global_step"] = torch.tensor(self.global_step, dtype=torch.int64)
        sd["dropout_step"] = torch.tensor(int(getattr(self.model, "_step", 0)), dtype=torch.int64)
        torch.save(sd, path)
        return path

    def train(self):
        log.infov("Training starts")
        avg_step_time, avg_train_report = [0], {k: [0] for k in self.model.report}
        for s in range(self.max_train_iter):
            if s % self.train_average_iter == 0:
                self.log_message(self.global_step, avg_train_report, avg_step_time, "train", True)
                avg_train_report = {k: [] for k in avg_train_report}
                avg_step_time = []
            if s % self.validation_step == 0:
                rep, times, summary = {k: [] for k in self.model.report}, [], None
                for i in range(self.val_average_iter):
                    step, summary, loss, report, dt = self.run_val_step(i == self.val_average_iter - 1)
                    for k in rep:
                        rep[k].append(report[k])
                    times.append(dt)
                self.add_summary(summary)
                self.log_message(self.global_step, rep, times, "val", False)
            step, summary, loss, report, dt = self.run_train_step(s % self.heavy_summary_step == 0)
            for k in avg_train_report:
                avg_train_report[k].append(report[k])
            avg_step_time.append(dt)
            self.add_summary(summary)
            if s % self.checkpoint_step == 0:
                log.infov("Saved checkpoint at {}".format(step))
                self.save_checkpoint()

    def log_message(self, step, avg_report, avg_step_time, split="train", is_train=True):
        step_time = np.array(avg_step_time, dtype=np.float32).mean()
        if step_time == 0:
            step_time = 0.001
        log_str = "[{:5s} step {:4d} ".format(split, step)
        log_str += "({:.3f} sec/batch, {:.3f} instances/sec)]\n".format(step_time, self.batch_size / step_time)
        for key in sorted(avg_report.keys()):
            log_str += "  * {}: {:.5f}\n".format(key, np.array(avg_report[key], dtype=np.float32).mean())
        if self.rank == 0:
            (log.info if is_train else log.infov)(log_str)
        return log_str


def str2bool(v):
    return str(v).lower() == "true"


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--data_dir", type=str, default="data/preprocessed/visualgenome"
                        "/memft_all_new_vocab50_obj3000_attr1000_maxlen10", help=" ")
    parser.add_argument("--image_dir", type=str, default="data/VisualGenome/VG_100K", help=" ")
    parser.add_argument("--max_train_iter", type=int, default=4810)
    parser.add_argument("--train_average_iter", type=int, default=10)
    parser.add_argument("--val_average_iter", type=int, default=40)
    parser.add_argument("--heavy_summary_step", type=int, default=200)
    parser.add_argument("--validation_step", type=int, default=200)
    parser.add_argument("--checkpoint_step", type=int, default=800)
    parser.add_argument("--prefix", type=str, default="default", help=" ")
    parser.add_argument("--checkpoint", type=str, default=None)
    parser.add_argument("--pretrained_param_path", type=str, default=None)
    parser.add_argument("--learning_rate", type=float, default=0.001, help=" ")
    parser.add_argument("--lr_weight_decay", action="store_true", default=False)
    parser.add_argument("--expand_depth", type=str2bool, default=False)
    parser.add_argument("--enwiki_preprocessing", type=int, default=0)
    parser.add_argument("--debug", type=int, default=0)
    parser.add_argument("--seed", type=int, default=123, help=" ")
    parser.add_argument("--batch_size", type=int, default=512, help=" ")
    parser.add_argument("--model_type", type=str, default="vlmap_bf_or_wordset_withatt_sp", help=" ",
                        choices=MODEL_TYPES)
    # not in the reference: how the input side keeps up with a 18 ms step
    parser.add_argument("--features_on_device", type=int, default=1,
                        help="keep both splits' feature tables in HBM, batches carry image indices")
    parser.add_argument("--ln_shared", type=int, default=1,
                        help="1: one LayerNorm per shared fc_layer scope (what TF 1.x builds); 0: one per call site; "
                             "a --checkpoint's variable names override this")
    parser.add_argument("--input_workers", type=int, default=4, help="forked batch producers (0: in-process)")
    parser.add_argument("--input_prefetch", type=int, default=2, help="batches assembled ahead of the step")
    return parser


def main(argv=None):
    config = build_parser().parse_args(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        config.device = "cuda:%d" % local
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        config.input_workers = 0      # ranks are already one process per GPU; producers are forked before the GPU is touched
    torch.manual_seed(config.seed)
    np.random.seed(config.seed)
    dataset = {"train": dataset_vlmap.Dataset(config, "train", seed=config.seed),
               "val": dataset_vlmap.Dataset(config, "val", seed=config.seed)}
    config.data_cfg = dataset["train"].get_config()
    Trainer(config, dataset).train()


if __name__ == "__main__":
    main()
