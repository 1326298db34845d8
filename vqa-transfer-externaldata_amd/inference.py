"""Counterpart of vqa/inference.py:1-110 -- the helper the reference's notebooks use to get a model restored from a
checkpoint together with its four input pipelines.

    from vqa_transfer_externaldata_amd import inference
    config = inference.get_default_config(); config.checkpoint = 'train_dir/vqa_<type>_d_<split>_tf_record_memft_.../model-4801'
    config.vlmap_word_weight_dir = '.../word_weights_model-N'
    inference.parse_checkpoint(config)                    # model_type, tf_record_dir, vfeat / vocab paths from the run's name
    inf = inference.get_inference(config)                 # inf.model (restored), inf.batches[split] iterators

`parse_checkpoint` follows the run-directory grammar of vqa/trainer.py:28-41 exactly as :79-95 does.  Where the reference
holds one tf.case over four pipelines and a session, this holds the iterators and a Model whose `set_batch` + `build` run a
batch (`Inference.run(split)`)."""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from . import importer, input_ops_vqa
from .log import log


class Inference(object):

    @staticmethod
    def get_model_class(model_type="vqa"):
        return importer.get_model_class(model_type)

    def __init__(self, config, datasets=None, image_features=None):
        self.config = config
        self.vfeat_path = config.vfeat_path
        self.tf_record_dir = config.tf_record_dir
        self.train_dir = os.path.dirname(config.checkpoint)
        if getattr(config, "vlmap_word_weight_dir", None) is not None:      # the copy the trainer left in the run directory (:19-22)
            self.vlmap_word_weight_dir = os.path.join(self.train_dir, config.vlmap_word_weight_dir.rstrip("/").split("/")[-1])
            config.vlmap_word_weight_dir = self.vlmap_word_weight_dir
        self.batch_size = config.batch_size
        ds = datasets or {}
        self.batches = {}
        for split, shuffle in (("train", True), ("val", False), ("testval", False), ("test", False)):
            if split in ds or os.path.exists(os.path.join(self.tf_record_dir, split + ".npz")) or \
                    os.path.isdir(os.path.join(self.tf_record_dir, split)):
                self.batches[split] = input_ops_vqa.create(self.batch_size, self.tf_record_dir, split, is_train=True,
                                                           scope="%s_ops" % split, shuffle=shuffle, data=ds.get(split))
        Model = self.get_model_class(config.model_type)
        log.infov("using model class: {}".format(Model))
        first = next(self.batches["train" if "train" in self.batches else sorted(self.batches)[0]])
        self.model = Model(first, config, is_train=True, image_features=image_features)
        self.ckpt_path = config.checkpoint
        if self.ckpt_path is not None:
            log.info("Checkpoint path: {}".format(self.ckpt_path))
            self.model.engine.load_state_dict(torch.load(self.ckpt_path, map_location="cpu"))
            log.info("Loaded the checkpoint")
        log.warning("Inference initialization is done")

    def run(self, split="val"):
        """one batch of `split` through the restored model: (batch, model) with model.output / mid_result filled"""
        batch = next(self.batches[split])
        self.model.set_batch(batch)
        self.model.build()
        return batch, self.model


def get_model_types():
    return importer.get_model_types()


def parse_checkpoint(config):
    """vqa/inference.py:79-95"""
    config.ckpt_name = config.checkpoint.split("/")[-1]
    dirname = config.checkpoint.split("/")[-2]
    config.model_type = dirname.split("vqa_")[1].split("_d_")[0]
    qa_split_name = dirname.split("_d_")[1].split("_tf_record_memft")[0]
    config.tf_record_dir = os.path.join("data/preprocessed/vqa_v2", qa_split_name, "tf_record_memft")
    config.vfeat_name = "vfeat_bottomup_36_my.hdf5" if "vfeat_bottomup_36_my" in dirname else "vfeat_bottomup_36.hdf5"
    config.vocab_path = os.path.join(config.tf_record_dir, config.vocab_name)
    config.vfeat_path = os.path.join(config.tf_record_dir, config.vfeat_name)


def get_default_config():
    """vqa/inference.py:98-104"""
    return SimpleNamespace(image_dir="data/VQA_v2/images", vocab_name="vocab.pkl", checkpoint=None, batch_size=512)


def get_inference(config=None, **kw):
    if config is None:
        config = get_default_config()
    return Inference(config, **kw)
