"""MI355X counterpart of vqa/model_vlmap_answer_no_noise.py: the deterministic half of the VAE variant.

`q_L_mean = fc_layer(q_L_ft, L_DIM, use_bias=True, no LayerNorm, no activation, scope='q_L_mean')` (:122-125) feeds
`q_linear_l` (:157); everything else is model_vlmap_answer.  One of the three oldest variants: no `output` /
`heavy_output`, the 9-key report of :210-219 plus `model_step` (:33).  `model_type` 8 of the C step."""
from .model_standard_testmask import REPORT_KEYS
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_no_noise"
    REPORT_RENAME = REPORT_KEYS

    def _constant_report(self):
        return {"model_step": int(self._engine.step_count) if self._engine is not None else 0}      # tf global_step (:32-33)

    def build(self):
        loss = super().build()
        B = int(self._db["q_intseq"].shape[0])
        self.mid_result["q_L_mean"] = self._engine.tensor("q_L_mean").view(B, -1)
        return loss
