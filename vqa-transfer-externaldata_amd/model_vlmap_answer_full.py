"""MI355X counterpart of vqa/model_vlmap_answer_full.py: the VAE variant of the question code.

Two linear heads on the GRU state, `q_L_mean` and `q_L_log_sigma_sq` (:124-131); `q_linear_l` reads the
reparameterised sample `q_L_mean + noise * sqrt(exp(q_L_log_sigma_sq))` (:132-134, 166); the loss adds
`latent_loss_weight (0.1) * latent_loss` with `latent_loss = -0.5 * mean_B sum (1 + ls - mean^2 - exp(ls))` (:217-223,
272-276).  `noise = tf.random_normal(seed=123)` is drawn inside the reference graph; here it is an explicit reproducible
input keyed by (config.seed, step) like the dropout masks (FusionEngine.make_noise -> vqa_normal_noise).  Report: the 9
older keys + latent_loss, train_latent_loss, model_step, latent_loss_weight (:34-35, 222-232).  `model_type` 10."""
from . import fusion as F
from .model_standard_testmask import REPORT_KEYS
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_full"
    REPORT_RENAME = REPORT_KEYS

    def _constant_report(self):
        return {"model_step": int(self._engine.step_count) if self._engine is not None else 0,
                "latent_loss_weight": F.LATENT_LOSS_WEIGHT}

    def _variant_inputs(self, eng, seed, row_offset, global_rows, dropout_off):
        return {"noise": eng.make_noise(seed, self._step, row_offset=row_offset, global_rows=global_rows)}

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        for k in ("q_L_mean", "q_L_log_sigma_sq", "q_L_mean_noise"):
            self.mid_result[k] = eng.tensor(k).view(B, -1)
        return loss
