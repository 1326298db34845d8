"""BASELINE config 5 end to end on tiny synthetic data: pre-train (cfg-5 model) -> export word weights ->
VQA model_vlmap_answer fine-tune with --pretrained_param_path + --vlmap_word_weight_dir -> evaluate."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pretrain_export_finetune_eval(tmp_path):
    from vqa_transfer_externaldata_amd import dataset_vlmap as DV
    from vqa_transfer_externaldata_amd import evaler, input_ops_vqa, pretrain_trainer as PTT, trainer
    Vq, n_ws, A_pre, R, D = 80, 20, 40, 36, 64
    data = DV.synthetic_dataset(24, Vq, n_ws, A_pre, R=R, D=D, max_len=8, seed=1)
    ds = {"train": DV.Dataset(split="train", data=data, seed=1), "val": DV.Dataset(split="val", data=data, seed=2)}
    b = next(DV.create_ops(8, ds["val"], is_train=False))
    assert b["obj_blank_fill/blanks"].shape[:2] == (8, 5) and b["image_ft"].shape == (8, R, D)
    assert b["obj_blank_fill/num"].min() >= 1 and b["attr_blank_fill/wordsets"].dtype == np.int32

    cfg = PTT.build_parser().parse_args(["--batch_size", "8", "--max_train_iter", "9", "--train_average_iter", "3",
                                         "--val_average_iter", "1", "--validation_step", "4", "--checkpoint_step", "4",
                                         "--heavy_summary_step", "4", "--learning_rate", "0.002",
                                         "--input_workers", "0"])      # (this process already holds the GPU)
    cfg.data_cfg = ds["train"].get_config()
    cfg.vocab = {"vocab": ["w%d" % i for i in range(Vq)], "dict": {"w%d" % i: i for i in range(Vq)}}
    cfg.answer_dict, cfg.ws_dict = data["answer_dict"], data["ws_dict"]
    cfg.synthetic, cfg.train_dir = 1, str(tmp_path / "pre")
    t = PTT.Trainer(cfg, ds)
    assert len(t.model.report) == 13 and "obj_wordset_top_5_acc" in t.model.report and "total_loss" in t.model.report
    l0 = t.model.report["total_loss"]
    t.train()
    ckpt = os.path.join(cfg.train_dir, "model-9")
    assert os.path.exists(ckpt) and t.model.report["total_loss"] < l0
    sd = torch.load(ckpt)
    wdir = PTT.export_word_weights(sd, cfg.vocab, cfg.answer_dict, os.path.join(cfg.train_dir, "word_weights_model-9"))

    # VQA fine-tune on answers that partly overlap the pre-training answer vocabulary
    vc = trainer.parse_config(["--batch_size", "8", "--max_train_iter", "5", "--train_average_iter", "2",
                               "--val_average_iter", "1", "--validation_step", "4", "--checkpoint_step", "4",
                               "--model_type", "vlmap_answer", "--pretrained_param_path", ckpt,
                               "--vlmap_word_weight_dir", wdir])
    A = 30
    vc.vocab = cfg.vocab
    vc.answer_dict = {"vocab": ["a%d" % (2 * i) for i in range(A)], "dict": {"a%d" % (2 * i): i for i in range(A)},
                      "num_train_answer": 20, "is_object": [i % 2 for i in range(A)],
                      "is_attribute": [1 - i % 2 for i in range(A)]}
    vc.synthetic, vc.train_dir, vc.tf_record_dir = 1, str(tmp_path / "vqa"), str(tmp_path / "none")
    feats = {"features": data["image_features"], "spatials": data["spatial_features"],
             "normal_boxes": data["normal_boxes"], "num_boxes": data["num_boxes"], "max_box_num": R, "vfeat_dim": D}
    splits = {"train": input_ops_vqa.synthetic_split(32, 24, Vq, A, seed=3),
              "testval": input_ops_vqa.synthetic_split(16, 24, Vq, A, seed=4)}
    vt = trainer.Trainer(vc, datasets=splits, image_features=feats)
    v = vt.model.variables()
    for n in ("pooled_linear_l/fc/weights", "q_linear_l/LayerNorm/gamma", "joint_fc/fc/biases"):
        assert torch.equal(v[n].cpu(), sd[n]), n                         # transferred from pre-training
    head = v["WordWeightAnswer/fc/weights"].cpu().numpy()
    cw = sd["classifier/fc/weights"].numpy()
    np.testing.assert_array_equal(head[:, 3], cw[:, 6])                   # answer 'a6' by string lookup
    assert np.all(head[:, 25] == 0) and float(v["WordWeightAnswer/fc/biases"][25]) == -100.0   # 'a50' unseen
    assert float(vt.model.answer_exist_mask.sum()) == 20                  # a0..a38 even exist in pre-training
    vt.train()
    ck2 = os.path.join(vc.train_dir, "model-5")
    ec = argparse.Namespace(**vars(vc))
    ec.checkpoint, ec.split, ec.max_iter, ec.dump_heavy_output = ck2, "testval", -1, False
    res = evaler.Evaler(ec, image_features=feats, data=splits["testval"]).eval()
    assert len(res["qid2result"]) == 16 and "normal_test_object_acc" in res["avg_eval_report"]
