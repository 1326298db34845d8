"""MI355X counterpart of vqa/model_vlmap_answer_noc.py ("no composition"): instead of joint_fc(pooled_linear_l * l_linear_l)
two separate branches -- joint_v on pooled_linear_l and joint_l on l_linear_l, each FC + LayerNorm + ReLU + dropout 0.5
(:177-188) -- with their own transferred heads WordWeightAnswerV / WordWeightAnswerL (weights.hdf5 datasets v_class_* /
l_class_* written by vlmap_memft/export_noc_word_weights.py:72-75; :190-202); logit = v_logit + l_logit (:204); loss, argmax
and report as in model_vlmap_answer.  Frozen: q_linear_l, pooled_linear_l, joint_v, joint_l and both heads (:80-90);
transferred: the four layers (:92-103).  `model_type` 5 of the C step (csrc/fusion_model.hip).  The pre-training variant
that produces v_class_* / l_class_* is a reference ablation outside this repo; without a word-weight directory both heads
are the untrained ones (weights 0, bias -100), as in the reference."""
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_noc"

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        self.mid_result.pop("joint", None)                                     # the reference has no composed `joint` here
        self.mid_result["v_joint"] = eng.tensor("joint").view(B, -1)
        self.mid_result["l_joint"] = eng.tensor("l_joint").view(B, -1)
        return loss
