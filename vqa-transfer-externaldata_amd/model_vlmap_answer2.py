"""MI355X counterpart of vqa/model_vlmap_answer2.py: model_vlmap_answer with one more question layer.

`q_L_ft2 = fc_layer(q_L_ft, V_DIM, LayerNorm, tanh, scope='q_L_ft2')` (:127-130) is what `q_linear_l` reads (:164) and
what `heavy_output['condition']` exposes (:131); `q_linear_v` -- the attention's query -- keeps reading the GRU state
(:135-138).  `q_L_ft2` is trainable (not in the frozen set of :69-75).  Report / output as in model_vlmap_answer
(:261-273).  `model_type` 7 of the C step (csrc/fusion_model.hip; tanh + LayerNorm = vqa_ln_act_fwd / _bwd).
The reference constructor is (batch, config, is_train) (:17); `image_features` stays an optional extra here."""
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer2"

    def build(self):
        loss = super().build()
        B = int(self._db["q_intseq"].shape[0])
        self.mid_result["q_L_ft2"] = self._engine.tensor("q_L_ft2").view(B, -1)
        return loss
