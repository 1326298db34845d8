"""MI355X counterpart of vqa/model_vlmap_only.py: model_vlmap_finetune with everything the pre-training produced kept
fixed -- V_WordMap, v_word_fc, q_linear_v, v_linear_v, hadamard_attention, q_linear_l, pooled_linear_l, joint_fc and the
WordWeightAnswer head are excluded from the train set (:64-76); what trains is the question side: LearnGloVe, the
bi-directional GRU, q_att_key, q_att_query and the word attention's score layer.  The two reference files are otherwise
identical; here the frozen variables simply have no gradient buffer (NULL members of `grads` in the C step)."""
from .model_vlmap_finetune import Model as _Finetune


class Model(_Finetune):
    MODEL_TYPE = "vlmap_only"
