"""MI355X counterpart of vqa/model_vlmap_finetune.py -- the bi-directional-GRU generation of the VQA model.

Differences from model_vlmap_answer (:119-190): the question is encoded by `modules.encode_L_bidirection` (two GRUCells
of 512 units, tf.nn.bidirectional_dynamic_rnn; vlmap/modules.py:100-122): `q_L_map` [B,T,1024] = both directions'
outputs, `q_L_ft` [B,1024] = both final states; a question SELF-attention (`q_att_key` = fc_layer(q_L_map) with its
LayerNorm over the [T,1024] block, `q_att_query` = fc_layer(q_L_ft), hadamard_attention under scope `word_attention`)
pools a second, transferred word embedding (`V_WordMap` -> `v_word_fc`) into `pooled_q_v`, and THAT is what
`q_linear_v` turns into the image attention's query; `q_linear_l` reads `q_L_ft`.  Every variable trains (:64-68);
`pretrained_param_path` and `vlmap_word_weight_dir` are mandatory (:23-28); report = answer_train_loss,
answer_report_loss, answer_accuracy (:207-211).  `model_type` 12 of the C step (csrc/fusion_model.hip:
bi_question_fwd / _bwd_*; csrc/bi_ops.hip; the recurrences are the fused GRU-step kernels with 512 columns).
The reference constructor is (batch, config, is_train); `image_features` stays an optional extra here."""
from .model_vlmap_answer import Model as _Base

REPORT_KEYS = (("answer_train_loss", "answer_train_loss"), ("answer_report_loss", "answer_report_loss"),
               ("answer_accuracy", "answer_acc"))


class Model(_Base):
    MODEL_TYPE = "vlmap_finetune"
    REPORT_RENAME = REPORT_KEYS

    def __init__(self, batch, config, is_train=True, image_features=None):
        if getattr(config, "pretrained_param_path", None) is None:
            raise ValueError("pretrained_param_path is mendatory")          # (sic) :24-25
        if getattr(config, "vlmap_word_weight_dir", None) is None:
            raise ValueError("word_weight_dir is mendatory")                # (sic) :27-28
        super().__init__(batch, config, is_train=is_train, image_features=image_features)

    def _variant_inputs(self, eng, seed, row_offset, global_rows, dropout_off):
        if dropout_off:
            return {}
        return {"keep_word": eng.make_keep_mask_word(seed, self._step, row_offset=row_offset, global_rows=global_rows)}

    def build(self):
        loss = super().build()
        eng, B = self._engine, int(self._db["q_intseq"].shape[0])
        T = eng.dims.T
        self.mid_result.update(q_L_map=eng.tensor("q_L_map").view(B, T, -1), q_L_ft=eng.tensor("q_L_ft").view(B, -1),
                               w_att_score=eng.tensor("w_att_score").view(B, T), pooled_q_v=eng.tensor("pooled_q_v").view(B, -1))
        return loss
