"""MI355X counterpart of vqa/model_vlmap_answer_ent.py: model_vlmap_answer + a maximum-entropy regulariser.

Every question is paired with NUM_MARGINAL (200, :16) pooled visual features of the batch -- `tile_pooled_linear_l =
reshape(tile(stop_gradient(pooled_linear_l), [200, 1]), [-1, 200, L])`, i.e. pairing (i, m) reads row (200 i + m) mod B
(:196-198) -- pushed through `joint_fc` (whose LayerNorm then normalises over the whole [200, 2048] block), its own
dropout and the WordWeightAnswer head (:199-207); the softmax over the known training answers (:63-65, 208-210) is averaged
over the pairings and `W_ENTROPY (0.1) * mean_B sum_a p log(p + 1e-8)` joins the loss (:281-292).  Only `l_linear_l`
receives a gradient from it.  At bs 512 that is 102 400 rows through two frozen GEMMs forward and backward (2.7 TFLOP,
about six times the base step) -- tile_pooled is never materialised, the head runs on the leading known-answer columns
only.  Report: the 13 keys + entropy, weighted_entropy (:293-307).  `model_type` 11 of the C step."""
from . import fusion as F
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_ent"

    def _engine_kwargs(self):
        return {"num_marginal": int(getattr(self.config, "num_marginal", F.NUM_MARGINAL))}

    def _variant_inputs(self, eng, seed, row_offset, global_rows, dropout_off):
        if dropout_off:
            return {}
        return {"keep_tile": eng.make_keep_mask_tile(seed, self._step, row_offset=row_offset, global_rows=global_rows)}

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        self.mid_result["marginal_prob"] = eng.tensor("marginal_prob").view(B, eng.dims.ent_cols)
        return loss
