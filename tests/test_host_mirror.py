"""CPU tests of the host-side mirror of the reference interface: CLI flags and
defaults, run-directory grammar, log line, model registry, batch provider contract."""
import argparse

import numpy as np
import pytest

from vqa_transfer_externaldata_amd import evaler, importer, input_ops_vqa, trainer


def test_trainer_flags_and_defaults_match_reference():
    # vqa/trainer.py:325-356
    c = trainer.parse_config([])
    want = dict(image_dir="data/VQA_v2/images", vfeat_name="vfeat_bottomup_36_my.hdf5", vocab_name="vocab.pkl",
                max_train_iter=7300, train_average_iter=200, val_average_iter=419, heavy_summary_step=800,
                validation_step=800, checkpoint_step=800, prefix="default", checkpoint=None,
                pretrained_param_path=None, learning_rate=0.001, lr_weight_decay=False, batch_size=512,
                model_type="vlmap_answer", vlmap_word_weight_dir=None, ft_vlmap=False, seed=123, debug=0)
    for k, v in want.items():
        assert getattr(c, k) == v, k
    assert c.tf_record_dir.endswith("thres1_50000_thres2_-1/tf_record_memft")
    assert c.vocab_path == c.tf_record_dir + "/vocab.pkl"
    assert c.vfeat_path == c.tf_record_dir + "/vfeat_bottomup_36_my.hdf5"


def test_check_config_rejects_checkpoint_plus_pretrained():
    with pytest.raises(ValueError, match="Do not set both"):
        trainer.parse_config(["--checkpoint", "a", "--pretrained_param_path", "b"])


def test_evaler_flags_and_parse_checkpoint():
    # vqa/evaler.py:195-232
    p = evaler.build_parser()
    with pytest.raises(SystemExit):
        p.parse_args([])                                    # --checkpoint is required
    c = p.parse_args(["--checkpoint",
                      "train_dir/vqa_vlmap_answer_d_qa_split_x_tf_record_memft_vfeat_bottomup_36_my_default_bs512_lr0.001"
                      "_seed123_20180101-000000/model-4801"])
    assert (c.split, c.max_iter, c.batch_size, c.dump_heavy_output) == ("testval", -1, 512, False)
    evaler.parse_checkpoint(c)
    assert c.ckpt_name == "model-4801" and c.model_type == "vlmap_answer"
    assert c.tf_record_dir == "data/preprocessed/vqa_v2/qa_split_x/tf_record_memft"
    assert c.vfeat_name == "vfeat_bottomup_36_my.hdf5"
    c2 = argparse.Namespace(checkpoint="t/vqa_standard_d_qa_s_tf_record_memft_vfeat_bottomup_36_default_bs512/model-1",
                            vocab_name="vocab.pkl")
    evaler.parse_checkpoint(c2)
    assert c2.model_type == "standard" and c2.vfeat_name == "vfeat_bottomup_36.hdf5"


def test_log_message_format():
    t = trainer.Trainer.__new__(trainer.Trainer)
    t.batch_size = 512
    s = t.log_message(800, {"b_acc": [0.25, 0.75], "a_loss": [2.0]}, [0.5, 0.5], split="train", is_train=True)
    assert s == ("[train step  800 (0.500 sec/batch, 1024.000 instances/sec)]\n"
                 "  * a_loss: 2.00000\n  * b_acc: 0.50000\n")
    s = t.log_message(1, {"x": [0]}, [0], split="val", is_train=False)      # zero step time -> 0.001
    assert "(0.001 sec/batch, 512000.000 instances/sec)" in s


def test_importer_registry():
    native = importer.get_model_types()
    assert native == ["standard", "standard_testmask", "standard_word2vec", "vlmap_answer", "vlmap_answer_noc",
                      "vlmap_answer_nocarch", "vlmap_answer_vqa_all", "vlmap_answer_vqa_all2", "vlmap_answer2",
                      "vlmap_answer_adapt", "vlmap_answer_ent", "vlmap_answer_full", "vlmap_answer_no_noise", "vlmap_finetune",
                      "vlmap_only", "vqa"]
    assert importer.get_model_class("vlmap_answer_") is importer.get_model_class("vlmap_answer_vqa_all")      # vqa/importer.py:33
    assert importer.get_model_class("vlmap_answer_nocarch").__mro__[1] is importer.get_model_class("vlmap_answer_noc")
    for t in native:
        assert importer.get_model_class(t).MODEL_TYPE == t
    # every entry of the reference registry (vqa/importer.py:1-14) is either native or refused by name
    reference = ["vqa", "standard", "standard_testmask", "standard_word2vec", "vlmap_only", "vlmap_finetune", "vlmap_answer",
                 "vlmap_answer_vqa_all", "vlmap_answer_vqa_all2", "vlmap_answer2", "vlmap_answer_noc", "vlmap_answer_nocarch",
                 "vlmap_answer_adapt", "vlmap_answer_ent", "vlmap_answer_full", "vlmap_answer_no_noise"]
    assert set(native) == set(reference)                                     # all 16 entries are native
    with pytest.raises(ValueError, match="Unknown model_type"):
        importer.get_model_class("nope")


def test_input_ops_batch_contract(tmp_path):
    d = input_ops_vqa.synthetic_split(70, 10, 50, 21, max_len=14, seed=3)
    d.save(str(tmp_path), "val")
    it = input_ops_vqa.create(32, str(tmp_path), "val", is_train=False, shuffle=False)
    batches = list(it)
    assert [len(b["id"]) for b in batches] == [32, 32, 6]                  # last batch short
    b = batches[0]
    assert b["id"].dtype == np.int64 and b["image_idx"].dtype == np.int64
    assert b["q_intseq"].dtype == np.int32 and b["q_intseq_len"].dtype == np.int32
    assert b["answer_target"].dtype == np.float32 and b["answer_target"].shape == (32, 21)
    assert b["q_intseq"].shape[1] == b["q_intseq_len"].max()               # padded to the batch max
    for i in range(32):
        L = b["q_intseq_len"][i]
        assert np.all(b["q_intseq"][i, L:] == 0) and np.all(b["q_intseq"][i, :L] > 0)
    assert set(np.unique(b["answer_target"])) <= {0.0, np.float32(0.3), np.float32(0.6), np.float32(0.9), 1.0}
    np.testing.assert_array_equal(np.concatenate([x["id"] for x in batches]), np.arange(70))


def test_train_pipeline_caches_batches_after_shuffle():
    d = input_ops_vqa.synthetic_split(50, 10, 50, 21, seed=4)
    it = input_ops_vqa.create(16, None, "train", is_train=True, shuffle=True, seed=7, data=d, repeat=3)
    batches = list(it)
    assert len(batches) == 3 * 4
    first = [b["id"] for b in batches[:4]]
    assert not np.array_equal(np.concatenate(first), np.arange(50))        # shuffled
    for e in (1, 2):                                                        # identical batches every epoch
        for i in range(4):
            np.testing.assert_array_equal(batches[4 * e + i]["id"], first[i])


def test_trainer_standard_flags_model_choices_and_clip():
    """vqa/trainer_standard.py: its own defaults (:282-300), only `standard` on the command line, standard /
    standard_word2vec through get_model_class (:17-24), gradient clip 0.25 (:95)"""
    from vqa_transfer_externaldata_amd import trainer_standard as TS
    c = TS.parse_config([])
    assert c.tf_record_dir.endswith("bottomup_vqa_tf_record_memft") and c.vfeat_name == "vfeat_bottomup_36.hdf5"
    assert c.model_type == "standard" and c.batch_size == 512 and c.checkpoint_step == 800 and c.val_average_iter == 419
    assert c.vfeat_path.endswith("vfeat_bottomup_36.hdf5") and c.vocab_path.endswith("vocab.pkl")
    with pytest.raises(SystemExit):
        TS.parse_config(["--model_type", "vlmap_answer"])
    assert TS.Trainer.get_model_class("standard").MODEL_TYPE == "standard"
    assert TS.Trainer.get_model_class("standard_word2vec").MODEL_TYPE == "standard_word2vec"
    with pytest.raises(ValueError, match="Unknown model_type"):
        TS.Trainer.get_model_class("vlmap_answer")
    assert TS.CLIP_GRADIENTS == 0.25 and TS.MAX_STEPS == 1000000


def test_inference_parse_checkpoint_follows_the_run_directory_grammar():
    """vqa/inference.py:79-104"""
    from vqa_transfer_externaldata_amd import inference as INF
    c = INF.get_default_config()
    assert (c.image_dir, c.vocab_name, c.checkpoint, c.batch_size) == ("data/VQA_v2/images", "vocab.pkl", None, 512)
    c.checkpoint = ("train_dir/vqa_vlmap_answer_vqa_all2_d_qa_split_objattr_thres_tf_record_memft_vfeat_bottomup_36_my_"
                    "default_bs512_lr0.001_seed123_20180101-000000/model-4801")
    INF.parse_checkpoint(c)
    # the reference splits on EVERY 'vqa_': a `vlmap_answer_vqa_all*` run parses as 'vlmap_answer_' -- which is why its importer
    # accepts that spelling (vqa/importer.py:33); reproduced
    assert c.ckpt_name == "model-4801" and c.model_type == "vlmap_answer_"
    assert importer.get_model_class(c.model_type).MODEL_TYPE == "vlmap_answer_vqa_all"
    assert c.tf_record_dir == "data/preprocessed/vqa_v2/qa_split_objattr_thres/tf_record_memft"
    assert c.vfeat_name == "vfeat_bottomup_36_my.hdf5" and c.vfeat_path.endswith("tf_record_memft/vfeat_bottomup_36_my.hdf5")
    c.checkpoint = "train_dir/vqa_standard_d_x_tf_record_memft_vfeat_bottomup_36_default_bs512/model-1"
    INF.parse_checkpoint(c)
    assert c.model_type == "standard" and c.vfeat_name == "vfeat_bottomup_36.hdf5"
    assert INF.get_model_types() == importer.get_model_types()
