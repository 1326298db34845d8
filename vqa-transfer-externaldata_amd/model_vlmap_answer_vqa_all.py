"""MI355X counterpart of vqa/model_vlmap_answer_vqa_all.py (also reachable as model_type 'vlmap_answer_' in the reference's
importer): model_vlmap_answer_vqa_all2 with three differences (the diff of the two reference files) --
the fixed WordWeightAnswer logits of answers the word-weight directory does not know are replaced by the ROW MINIMUM of
the fixed logits (:192-194); the tuned loss term is taken on the summed logits, ce(logit + tuned_logit) (:236-237), and
both terms are masked by the train-answer mask in the training loss (:241-242); pred = argmax(logit + tuned_logit)
(:244).  `model_type` 6 of the C step (vqa_rowmin_mask_fwd / _bwd, vqa_loss2_fwd(sum_mode=1))."""
from .model_vlmap_answer_vqa_all2 import Model as _Base


class Model(_Base):
    MODEL_TYPE = "vlmap_answer_vqa_all"

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        self.mid_result["logit_raw"] = eng.tensor("logit_raw").view(B, eng.dims.A)       # before the substitution
        return loss
