"""Model class of the cfg-5 pre-training stage with the constructor / attributes of
vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:13-94: Model(batch, config, is_train) ->
.loss, .losses, .report (13 scalars), .mid_result, .vocab, .answer_dict, .ws_dict; config carries
data_cfg (n_obj_bf, n_attr_bf, max_box_num, vfeat_dim), data_dir, expand_depth.  build() runs the
forward pass of the current batch on libvqahot.so through pretrain.PretrainEngine."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import pretrain as PT
from .dataset_vlmap import _load_pickle

TOP_K = 5
W_DIM = 300   # Word dimension
L_DIM = 1024  # Language dimension
V_DIM = 1024


class Model(object):

    def __init__(self, batch, config, is_train=True):
        self.batch = batch
        self.config = config
        self.data_cfg = config.data_cfg
        self.data_dir = getattr(config, "data_dir", None)
        self.is_train = is_train
        self.device = torch.device(getattr(config, "device", "cuda:0"))
        self.losses, self.report, self.mid_result, self.vis_image = {}, {}, {}, {}

        self.vocab = getattr(config, "vocab", None) or _load_pickle(os.path.join(self.data_dir, "vocab.pkl"))
        self.answer_dict = getattr(config, "answer_dict", None) or _load_pickle(
            os.path.join(self.data_dir, "answer_dict.pkl"))
        self.num_answer = len(self.answer_dict["vocab"])
        self.ws_dict = getattr(config, "ws_dict", None) or _load_pickle(os.path.join(
            self.data_dir, "wordset_dict5_depth{}.pkl".format(int(getattr(config, "expand_depth", 0)))))
        self.num_ws = len(self.ws_dict["vocab"])
        self._step = 0
        self._engine = None
        self.build()

    def filter_train_vars(self, trainable_vars):
        return list(trainable_vars)                       # every variable trains (:45-51)

    def _initial_params(self, shapes):
        seed = int(getattr(self.config, "seed", 123))
        g = torch.Generator().manual_seed(seed)
        p = {}
        for n, s in shapes.items():
            if n.endswith("/weights") or n.endswith("/kernel"):
                lim = (6.0 / (s[0] + s[1])) ** 0.5
                p[n] = ((torch.rand(s, generator=g) * 2 - 1) * lim).numpy()
            elif n.endswith("gates/bias") or n.endswith("/gamma"):
                p[n] = np.ones(s, np.float32)
            elif n == "wordset_map/learn":                 # random_uniform(-0.01, 0.01), modules.py:351-358
                p[n] = (torch.rand(s, generator=g) * 0.02 - 0.01).numpy()
            elif n.endswith("embed_map"):                  # GloVe rows when available, else zeros (OOV)
                glove = getattr(self.config, "glove", None)
                p[n] = np.zeros(s, np.float32) if glove is None else glove[n]
                if glove is None and getattr(self.config, "synthetic", 0):
                    p[n] = (torch.rand(s, generator=g) * 0.02 - 0.01).numpy()
            else:
                p[n] = np.zeros(s, np.float32)
        return p

    @property
    def engine(self):
        return self._engine

    def variables(self):
        return dict(self._engine.params)

    def set_batch(self, batch):
        self.batch = batch

    def _device_batch(self):
        keep = ("image_ft", "spatial_ft", "num_boxes", "image_idx")
        out = {}
        for k, v in self.batch.items():
            if k in keep or k.split("/")[-1] in ("normal_boxes", "fills", "blanks", "blanks_len", "wordsets", "num"):
                if k == "normal_boxes":
                    continue
                t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
                out[k] = t.to(self.device)
        for k in ("image_ft", "spatial_ft"):
            if k in out:
                out[k] = out[k].float()
        if getattr(self.config, "sort_by_length", 1):
            host = PT.add_length_sort({k: v for k, v in self.batch.items() if k.endswith(("/blanks", "/blanks_len"))})
            for k, v in host.items():
                if k.endswith("/sort"):       # permutation / inverse uploaded here (on the stream of prepare())
                    v["_dev"] = (torch.from_numpy(np.asarray(v["perm"])).to(self.device, torch.int32),
                                 torch.from_numpy(np.asarray(v["inv"])).to(self.device, torch.int32),
                                 np.ascontiguousarray(v["live_rows"], dtype=np.int32))
                    out[k] = v
        for k in list(out):
            if k.endswith("_blank_fill/normal_boxes"):
                out[k] = out[k].float()
        if "_dp" in self.batch:               # data parallel: the shard's place in the global batch (Trainer._shard)
            out["_dp"] = self.batch["_dp"]
        return out

    def prepare(self, batch):
        """Host side of a step for `batch`, ahead of time: device copies (on a side stream, so they do not queue behind
        the step the GPU is running) and the length sort.  Returns a handle for build(prepared=...).  The trainer
        calls this for batch i+1 right after it has queued step i (the reference's tf.data pipeline prefetches the
        same way, vlmap_memft/datasets/dataset_vlmap.py:308-353)."""
        self.batch = batch
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._side):
            db = self._device_batch()
        ev = torch.cuda.Event()
        ev.record(self._side)
        return db, ev

    def build(self, prepared=None, defer_report=False):
        """build network architecture and loss (here: run it on self.batch, or on the batch handed to prepare()).
        defer_report: queue the forward only; finish_report() then fetches the 13 scalars (one host sync per step,
        after backward and the update have been queued as well)."""
        if prepared is None:
            db = self._device_batch()
        else:
            db, ev = prepared
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # the tensors were allocated on the side stream: tell the caching allocator that the compute stream uses them
            # too, so that their blocks are not handed out again (to the next prepare()) while this step still reads them
            for v in db.values():
                for t in (v.get("_dev", ()) if isinstance(v, dict) else (v,)):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
        if self._engine is None:
            cfg = self.data_cfg
            # one LayerNorm per shared fc_layer scope (TF 1.x: pretrain.py) unless the config asks for the
            # per-call-site variable set; a checkpoint's variable names override either (PretrainEngine.load_state_dict)
            shapes = PT.variable_shapes(len(self.vocab["vocab"]), self.num_ws, self.num_answer, W_DIM, cfg.vfeat_dim,
                                        V_DIM, bool(getattr(self.config, "ln_shared", 1)))
            self._engine = PT.PretrainEngine(n=cfg.n_obj_bf, R=cfg.max_box_num, D=cfg.vfeat_dim, H=V_DIM, W=W_DIM,
                                             A=self.num_answer, Vq=len(self.vocab["vocab"]), n_ws=self.num_ws,
                                             params=self._initial_params(shapes), device=self.device,
                                             deterministic=bool(getattr(self.config, "deterministic", 0)))
        eng = self._engine
        B = int((db["image_ft"] if "image_ft" in db else db["image_idx"]).shape[0])
        tables = getattr(self.config, "feature_tables", None)
        if tables is not None and getattr(eng, "_tables", None) is None:
            eng.bind_tables(*tables)           # (image_features, spatial_features, num_boxes): gathered on the device
        dpi = db.get("_dp") or {}
        masks = None if getattr(self.config, "dropout_off", False) else eng.make_keep_masks(
            B, int(getattr(self.config, "seed", 123)), self._step, row_offset=dpi.get("row_offset", 0),
            global_rows=dpi.get("global_rows"))
        self._step += 1
        self._reduce_report = bool(dpi)
        eng.forward(db, masks, global_valid=dpi.get("global_valid"))
        for k in PT.KINDS:
            kt = eng._tape["kinds"][k]
            name = "object" if k == "obj" else "attribute"
            self.mid_result[name + "_pooled_V_ft"] = kt["pooled"].view(B, eng.n, -1)
            self.mid_result[k + "_blank_fill/logit"] = kt["wordset"]["z"].view(B, eng.n, -1)
        if defer_report and self._reduce_report:
            self._report_event = None             # data parallel: the scalars are reduced over the ranks in finish_report
            return None
        if defer_report:
            if getattr(self, "_report_host", None) is None:
                self._report_host = torch.empty(16, dtype=torch.float32).pin_memory()
            self._report_host.copy_(eng.tensor("report")[:16], non_blocking=True)      # stream-ordered after the forward
            self._report_event = torch.cuda.Event()
            self._report_event.record(torch.cuda.current_stream(self.device))
            return None
        return self.finish_report()

    def finish_report(self):
        eng = self._engine
        if getattr(self, "_report_event", None) is not None:
            self._report_event.synchronize()
            r = self._report_host.numpy()
            self.report = eng.report = {eng.lib.vqa_pretrain_report_key(i).decode(): float(r[i]) for i in range(13)}
            self._report_event = None
        else:
            self.report = eng.fetch_report(reduce=getattr(self, "_reduce_report", False))
        self.losses = {k[:-5]: v for k, v in self.report.items() if k.endswith("_loss") and k != "total_loss"}
        self.loss = self.report["total_loss"]
        return self.loss

    def backward(self, reducer=None):
        self._engine.backward(reducer=reducer)

    def apply_gradients(self, learning_rate):
        self._engine.optimizer_step(learning_rate)
