"""MI355X counterpart of vqa/trainer.py: same flags and defaults (vqa/trainer.py:321-361),
train-dir name grammar (:28-41), loop cadence (:188-263), run_train_step / run_val_step return
tuples (:275-300), log line (:302-313) and checkpoint contract (model-<step> every
checkpoint_step; --checkpoint restores everything, --pretrained_param_path only the transfer
variables; both at once is an error, :316-318).  session.run is replaced by
Model.build() + backward() + apply_gradients() on libvqahot.so; under torch.distributed
(one process per GPU) the minibatch is sharded by sample and gradients are all-reduced over RCCL.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import time

import numpy as np
import torch

from . import dp, importer
from . import input_ops_vqa
from .log import log


class Trainer(object):

    @staticmethod
    def get_model_class(model_type="vlmap_answer"):
        return importer.get_model_class(model_type)

    def __init__(self, config, datasets=None, image_features=None):
        """datasets: optional {'train'|'val'|'testval'|'test': SplitData} overriding the files
        under config.tf_record_dir (used for synthetic runs and tests)."""
        self.config = config
        self.vfeat_path = config.vfeat_path
        self.tf_record_dir = config.tf_record_dir
        self.max_train_iter = config.max_train_iter

        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0

        dataset_str = "d"
        dataset_str += "_" + "_".join(config.tf_record_dir.replace("data/preprocessed/vqa_v2/", "").split("/"))
        dataset_str += "_" + config.vfeat_name.replace(".hdf5", "")
        hyper_parameter_str = "bs{}_lr{}".format(config.batch_size, config.learning_rate)
        if config.ft_vlmap:
            hyper_parameter_str += "_ft_vlmap"
        self.train_dir = getattr(config, "train_dir", None) or "./train_dir/vqa_{}_{}_{}_{}_seed{}_{}".format(
            config.model_type, dataset_str, config.prefix, hyper_parameter_str, config.seed,
            time.strftime("%Y%m%d-%H%M%S"))
        if self.rank == 0 and not os.path.exists(self.train_dir):
            os.makedirs(self.train_dir)
        log.infov("Train Dir: %s", self.train_dir)

        if config.vlmap_word_weight_dir is not None and os.path.isdir(config.vlmap_word_weight_dir):
            dst = os.path.join(self.train_dir, config.vlmap_word_weight_dir.rstrip("/").split("/")[-1])
            if self.rank == 0 and not os.path.exists(dst):
                shutil.copytree(config.vlmap_word_weight_dir, dst)
            if self.world > 1:
                torch.distributed.barrier()      # the other ranks read the copy (Model -> load_word_weight_dir)
            self.vlmap_word_weight_dir = config.vlmap_word_weight_dir = dst
        else:
            self.vlmap_word_weight_dir = config.vlmap_word_weight_dir

        # Input: four split pipelines, selected by name at run time (tf.case on a string, :72-79)
        self.batch_size = config.batch_size
        ds = datasets or {}
        self._iters = {}
        for split, shuffle in (("train", True), ("val", False), ("testval", False), ("test", False)):
            if split in ds or os.path.exists(os.path.join(self.tf_record_dir, split + ".npz")) or \
                    os.path.isdir(os.path.join(self.tf_record_dir, split)):
                self._iters[split] = input_ops_vqa.create(
                    self.batch_size, self.tf_record_dir, split, is_train=True, scope="%s_ops" % split,
                    shuffle=shuffle, seed=config.seed, data=ds.get(split))
        if "train" not in self._iters:
            raise ValueError("no training data under %s" % self.tf_record_dir)

        # Model
        Model = self.get_model_class(config.model_type)
        log.infov("using model class: {}".format(Model))
        config.global_batch = None
        first = self._shard(next(self._iters["train"]))
        if getattr(config, "sort_by_length", 1):
            first = (input_ops_vqa.sort_by_length(first[0]), first[1])
        self._dev_batches, self._dev_batch_bytes = {}, 0
        self._dev_batch_budget = int(getattr(config, "device_batch_cache_gb", 32)) << 30
        self._pending_train_batch, self._pending_global = first[0], None
        self._set_global(first[1])
        self.model = Model(first[0], config, is_train=True, image_features=image_features)

        # Optimizer (tf.contrib.layers.optimize_loss: Adam, clip 20.0, train_vars; :87-114)
        self.global_step = 0
        self.learning_rate = config.learning_rate
        all_vars = sorted(self.model.variables())
        self.train_vars = self.model.filter_train_vars(all_vars)
        self.transfer_vars = self.model.filter_transfer_vars(all_vars)
        log.warning("Filtered train variables: %s", ", ".join(self.train_vars))
        self._allreduce = dp.BucketedAllReduce() if self.world > 1 else None

        self.train_average_iter = config.train_average_iter
        self.val_average_iter = config.val_average_iter
        self.heavy_summary_step = config.heavy_summary_step
        self.validation_step = config.validation_step
        self.checkpoint_step = config.checkpoint_step
        self._summary_path = os.path.join(self.train_dir, "summaries.jsonl")

        self.ckpt_path = config.checkpoint
        if self.ckpt_path is not None:
            log.info("Checkpoint path: {}".format(self.ckpt_path))
            sd = torch.load(self.ckpt_path, map_location="cpu")
            self.model.engine.load_state_dict(sd)
            self.global_step = int(sd.get("global_step", 0))
            self.model._step = int(sd.get("dropout_step", self.global_step))   # dropout mask stream position
            log.info("Loaded the checkpoint")
        self.pretrained_param_path = config.pretrained_param_path
        if self.pretrained_param_path is not None:
            log.warning("Filtered transfer_vars (loaded from pre-trained param): %s", ", ".join(self.transfer_vars))
            sd = torch.load(self.pretrained_param_path, map_location="cpu")
            self.model.engine.load_state_dict(sd, var_names=self.transfer_vars)
            log.info("Loaded the pre-trained parameters")

    # ------------------------------------------------------------------ helpers
    def _shard(self, batch):
        if self.world == 1:
            return batch, len(batch["image_idx"])
        return dp.shard_batch(batch, self.rank, self.world)

    def _set_global(self, n_global):
        """global batch size of the batch about to run + the first global row of this rank's shard (it indexes
        the dropout mask stream, FusionEngine.make_keep_masks)"""
        self.config.global_batch = n_global
        self.config.shard_row_offset = (dp.shard_bounds(n_global, self.rank, self.world)[0]
                                        if self.world > 1 and n_global else 0)

    def _lr(self):
        if self.config.lr_weight_decay:     # tf.train.exponential_decay(staircase, 10000, 0.5)  (:89-96)
            return self.learning_rate * (0.5 ** (self.global_step // 10000))
        return self.learning_rate

    def _next(self, split):
        if split == "train" and self._pending_train_batch is not None:
            b, self._pending_train_batch = self._pending_train_batch, None
            if self._pending_global is not None:
                self._set_global(self._pending_global)
                self._pending_global = None
            return b
        it = self._iters.get(split) or self._iters["train"]
        raw = next(it)
        # The input pipeline caches its padded batches (dataset.cache() after padded_batch in the reference) and
        # yields the same dict objects every epoch, so this rank's shard of each batch is kept ON THE DEVICE
        # after its first use: 6.2 MB per bs-512 batch, ~5 GB for all of VQA v2 train against 288 GB of HBM --
        # from the second epoch on a step needs no host assembly and no host-to-device copy at all.
        hit = self._dev_batches.get(id(raw))
        if hit is not None and hit[0] is raw:
            batch, n_global = hit[1], hit[2]
        else:
            batch, n_global = self._shard(raw)
            if getattr(self.config, "sort_by_length", 1):
                # longest question first: every GRU step then runs on the still-running rows only
                batch = input_ops_vqa.sort_by_length(batch)
            if self._dev_batch_bytes < self._dev_batch_budget:
                batch = self.model.to_device_batch(batch)
                self._dev_batches[id(raw)] = (raw, batch, n_global)
                self._dev_batch_bytes += sum(v.numel() * v.element_size() for v in batch.values() if torch.is_tensor(v))
        self._set_global(n_global)
        return batch

    def _report_values(self):
        torch.cuda.synchronize(self.model.device)
        # data parallel: statistics are reduced over the ranks, so every rank logs the global-batch report
        gr = self.config.global_batch if self.world > 1 else None
        rep = self.model.engine.report(global_rows=gr)
        extra = self.model.engine.extra_report(global_rows=gr)      # latent_loss / entropy of the two variants that have one
        return float(extra.get("total_loss", rep["answer_train_loss"])), self.model.map_report(rep, extra)

    # ------------------------------------------------------------------ steps
    def run_train_step(self, use_heavy_summary):
        _start_time = time.time()
        self.model.set_batch(self._next("train"))
        self.model.build()
        self.model.backward(reducer=self._allreduce)       # buckets reduced while backward still runs
        self.model.apply_gradients(self._lr())
        # everything above is only ENQUEUED: assemble (and, first epoch, upload) the next batch while the GPU
        # works, and only then wait for this step's report
        gb = self.config.global_batch
        nxt = self._next("train")
        self._pending_train_batch, self._pending_global = nxt, self.config.global_batch
        self._set_global(gb)
        loss, report = self._report_values()
        self.global_step += 1
        _end_time = time.time()
        summary = {"split": "train", "step": self.global_step, **report} if use_heavy_summary else None
        return self.global_step, summary, loss, report, (_end_time - _start_time)

    def run_val_step(self, use_heavy_summary, split):
        _start_time = time.time()
        self.model.set_batch(self._next(split))
        self.model.build()
        loss, report = self._report_values()
        _end_time = time.time()
        summary = {"split": split, "step": self.global_step, **report} if use_heavy_summary else None
        return self.global_step, summary, loss, report, (_end_time - _start_time)

    def add_summary(self, summary):
        if summary is not None and self.rank == 0:
            with open(self._summary_path, "a") as f:
                f.write(json.dumps(summary) + "\n")

    def write_average_summary(self, window, split="train"):
        return self.global_step, dict({"split": "average_" + split, "step": self.global_step}, **window.means())

    def save_checkpoint(self):
        path = os.path.join(self.train_dir, "model-{}".format(self.global_step))
        if self.rank == 0:
            sd = self.model.engine.state_dict()
            sd["global_step"] = torch.tensor(self.global_step, dtype=torch.int64)
            sd["dropout_step"] = torch.tensor(int(self.model._step), dtype=torch.int64)
            torch.save(sd, path)
        return path

    def _validate(self, split):
        """val_average_iter inference steps on `split`, their running averages logged and summarised (:208-229)"""
        window, last = _Window(self.model.report), None
        for i in range(self.val_average_iter):
            _, last, _, report, seconds = self.run_val_step(i + 1 == self.val_average_iter, split=split)
            window.add(report, seconds)
        self.add_summary(last)
        step, averaged = self.write_average_summary(window, split=split)
        self.add_summary(averaged)
        self.log_message(step, window.report, window.seconds, split=split, is_train=False)

    def train(self):
        """The schedule of vqa/trainer.py:188-263: every train_average_iter steps the running averages are logged and
        reset, every validation_step steps the validation splits are sampled, heavy summaries every heavy_summary_step,
        a checkpoint every checkpoint_step (step 0 included)."""
        log.infov("Training starts")
        window = _Window(self.model.report, seed=0)         # the reference's first log line averages a single 0
        for s in range(self.max_train_iter):
            if s % self.train_average_iter == 0:
                step, averaged = self.write_average_summary(window, split="train")
                self.add_summary(averaged)
                self.log_message(step, window.report, window.seconds, split="train", is_train=True)
                window = _Window(self.model.report)
            if s % self.validation_step == 0:
                for split in ("val", "testval"):
                    if split in self._iters:
                        self._validate(split)
            heavy = s % self.heavy_summary_step == 0
            step, summary, _, report, seconds = self.run_train_step(heavy)
            window.add(report, seconds)
            if heavy:
                self.add_summary(summary)
            if s % self.checkpoint_step == 0:
                log.infov("Saved checkpoint at {}".format(step))
                self.save_checkpoint()

    def log_message(self, step, avg_report, avg_step_time, split="train", is_train=True):
        """The reference's log block (vqa/trainer.py:302-313), character for character: header with seconds per batch and
        instances per second, then one `  * key: value` line per report key in sorted order."""
        mean32 = lambda xs: np.asarray(xs, dtype=np.float32).mean()
        per_batch = mean32(avg_step_time) or 0.001
        lines = ["[{:5s} step {:4d} ({:.3f} sec/batch, {:.3f} instances/sec)]".format(
            split, step, per_batch, self.batch_size / per_batch)]
        lines += ["  * {}: {:.5f}".format(k, mean32(avg_report[k])) for k in sorted(avg_report)]
        text = "\n".join(lines) + "\n"
        (log.info if is_train else log.infov)(text)
        return text


class _Window:
    """Per-key lists of the report scalars and step times since the last log line."""

    def __init__(self, keys, seed=None):
        self.report = {k: ([] if seed is None else [seed]) for k in keys}
        self.seconds = [] if seed is None else [seed]

    def add(self, report, seconds):
        for k in self.report:
            self.report[k].append(report[k])
        self.seconds.append(seconds)

    def means(self):
        return {k: float(np.asarray(v, dtype=np.float32).mean()) for k, v in self.report.items()}


def check_config(config):
    if config.checkpoint is not None and config.pretrained_param_path is not None:
        raise ValueError("Do not set both checkpoint and pretrained_param_path")


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    # paths
    parser.add_argument("--image_dir", type=str, default="data/VQA_v2/images", help=" ")
    parser.add_argument("--tf_record_dir", type=str,
                        default="data/preprocessed/vqa_v2"
                        "/qa_split_objattr_answer_3div4_genome_memft_check_all_answer_thres1_50000_thres2_-1"
                        "/tf_record_memft", help=" ")
    parser.add_argument("--vfeat_name", type=str, default="vfeat_bottomup_36_my.hdf5", help=" ")
    parser.add_argument("--vocab_name", type=str, default="vocab.pkl", help=" ")
    # log
    parser.add_argument("--max_train_iter", type=int, default=7300)
    parser.add_argument("--train_average_iter", type=int, default=200)
    parser.add_argument("--val_average_iter", type=int, default=419)  # 419 for 1 epoch
    parser.add_argument("--heavy_summary_step", type=int, default=800)  # 867 for 1 epoch
    parser.add_argument("--validation_step", type=int, default=800)
    parser.add_argument("--checkpoint_step", type=int, default=800)
    # hyper parameters
    parser.add_argument("--prefix", type=str, default="default", help=" ")
    parser.add_argument("--checkpoint", type=str, default=None)
    parser.add_argument("--pretrained_param_path", type=str, default=None)
    parser.add_argument("--learning_rate", type=float, default=0.001, help=" ")
    parser.add_argument("--lr_weight_decay", action="store_true", default=False)
    # model parameters
    parser.add_argument("--batch_size", type=int, default=512, help=" ")
    parser.add_argument("--model_type", type=str, default="vlmap_answer", help=" ",
                        choices=importer.get_model_types())
    # model specific parameters
    parser.add_argument("--vlmap_word_weight_dir", type=str, default=None, help=" ")
    parser.add_argument("--ft_vlmap", action="store_true", default=False)
    parser.add_argument("--seed", type=int, default=123, help=" ")
    parser.add_argument("--sort_by_length", type=int, default=1,
                        help="order every batch by question length so the GRU skips finished sequences (not in the "
                             "reference; results are unchanged)")
    parser.add_argument("--device_batch_cache_gb", type=int, default=32,
                        help="keep the cached padded batches on the device up to this many GiB (not in the reference)")
    parser.add_argument("--debug", type=int, default=0, help="0: normal, 1: debug")
    return parser


def parse_config(argv=None):
    config = build_parser().parse_args(argv)
    config.vocab_path = os.path.join(config.tf_record_dir, config.vocab_name)
    config.vfeat_path = os.path.join(config.tf_record_dir, config.vfeat_name)
    check_config(config)
    return config


def main(argv=None):
    config = parse_config(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
        config.device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    torch.manual_seed(config.seed)
    np.random.seed(config.seed)
    trainer = Trainer(config)
    trainer.train()


if __name__ == "__main__":
    main()
