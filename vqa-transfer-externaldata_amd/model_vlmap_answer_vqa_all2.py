"""MI355X counterpart of vqa/model_vlmap_answer_vqa_all2.py -- the one model variant a launcher of the reference still
runs (run_vqa_all_non_standard.py:95).

model_vlmap_answer (transferred, frozen pooled_linear_l / q_linear_l / joint_fc and the WordWeightAnswer head with
default_bias -100: :128-196, frozen set :85-94, transfer set :96-105) plus a TRAINABLE second head
`TunedWordWeightAnswer` = fc_layer(joint, num_answer, use_bias=True) (:216-220).  The two logits are summed for
`output['logit']` (:226-227); the training loss is ce(fixed) * train_mask + ce(tuned), the report loss ce(fixed) +
ce(tuned) (:234-242); the prediction is argmax(fixed * test_mask + tuned * train_mask) (:243-244) -- the tuned head
answers with training answers, the transferred one with test answers.  All of it runs in the C step as `model_type` 4
(csrc/fusion_model.hip, vqa_loss2_fwd).

Reference quirk reproduced: `tuned_q_linear_l` and `tuned_joint_fc` (:202-214) are built, but the tuned head reads
`joint`, not `tuned_joint` (:216-217), so they feed nothing.  Their variables exist (checkpoint names), sit in the train
set and never receive a gradient; `tuned_mid_results()` evaluates the dead branch on request for `mid_result`."""
import numpy as np
import torch

from . import ops
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_vqa_all2"

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        A = eng.dims.A
        self.mid_result["logit_fixed"] = eng.tensor("logit_fixed").view(B, A)      # WordWeightAnswer (:191-195)
        self.mid_result["logit_tuned"] = eng.tensor("logit_tuned").view(B, A)      # TunedWordWeightAnswer (:216-220)
        return loss

    def tuned_mid_results(self, keep_mask=None):
        """mid_result['tuned_l_linear_l'] / ['tuned_joint'] of :202-214 for the batch of the last build().  keep_mask:
        uint8 [B, 2H] dropout keep-mask of tuned_joint (tf.nn.dropout(tuned_joint, 0.5)); None = no dropout."""
        eng = self._engine
        B, H = int(self._db["q_intseq"].shape[0]), eng.dims.H
        P = eng.params
        h = eng.tensor("condition").view(B, H)

        def fc_ln_relu(x, scope, keep=None):
            pre = ops.gemm(x, P[scope + "/fc/weights"], bias=P[scope + "/fc/biases"])
            return ops.ln_act_fwd(pre, P[scope + "/LayerNorm/gamma"], P[scope + "/LayerNorm/beta"], rows=1, act="relu",
                                  keepmask=keep, keep_prob=0.5 if keep is not None else 1.0)[0]

        tll = fc_ln_relu(h.contiguous(), "tuned_q_linear_l")
        tj = fc_ln_relu((eng.tensor("pooled_linear_l").view(B, H) * tll).contiguous(), "tuned_joint_fc", keep_mask)
        self.mid_result["tuned_l_linear_l"], self.mid_result["tuned_joint"] = tll, tj
        return tll, tj
