"""MI355X counterpart of vqa/model_vlmap_answer_adapt.py: the attention pools an ADAPTED visual feature.

`v_adapt = fc_layer(V_ft, V_DIM, LayerNorm, ReLU, scope='v_adapt')` (:132-135; like v_linear_v, its LayerNorm runs over the
whole [36, 1024] block of a sample) and `pooled_V_ft = attention_pooling(v_adapt, att_score)` (:142): the pooled vector is
1024-wide, so `pooled_linear_l/fc/weights` is [1024, 1024] here (a pre-trained [2048, 1024] one cannot be transferred --
as in the reference, whose restore would fail on the shape).  v_adapt is trainable, so the step carries a second
77-GFLOP weight-gradient GEMM.  9-key report (:212-220).  `model_type` 9 of the C step."""
from .model_standard_testmask import REPORT_KEYS
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_adapt"
    REPORT_RENAME = REPORT_KEYS

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        self.mid_result["v_adapt"] = eng.tensor("v_adapt").view(B, eng.dims.R, eng.dims.H)
        return loss
