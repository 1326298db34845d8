"""String -> Model class registry, the counterpart of vqa/importer.py:1-52.

The two models on the hot path and the variants of section 8f-4 -- standard_word2vec,
standard_testmask, vlmap_answer_vqa_all, vlmap_answer_vqa_all2, vlmap_answer_noc = vlmap_answer_nocarch and the five older
ablations vlmap_answer2 / _no_noise / _adapt / _full / _ent, the two bi-directional-GRU models vlmap_finetune / vlmap_only
and the oldest model `vqa` (an LSTM encoder over the 512-d features of the model_vfeat pipeline that scores every
answer's own LSTM code) -- are built natively: all 16 entries of vqa/importer.py:1-14."""

_NATIVE = ("standard", "standard_testmask", "standard_word2vec", "vlmap_answer", "vlmap_answer_noc", "vlmap_answer_nocarch",
           "vlmap_answer_vqa_all", "vlmap_answer_vqa_all2", "vlmap_answer2", "vlmap_answer_adapt", "vlmap_answer_ent",
           "vlmap_answer_full", "vlmap_answer_no_noise", "vlmap_finetune", "vlmap_only", "vqa")
_REFERENCE_ONLY = ()


def get_model_types():
    return list(_NATIVE)


def get_model_class(model_type="vlmap_answer"):
    if model_type == "standard":
        from .model_standard import Model
    elif model_type == "standard_testmask":
        from .model_standard_testmask import Model
    elif model_type == "standard_word2vec":
        from .model_standard_word2vec import Model
    elif model_type == "vlmap_answer":
        from .model_vlmap_answer import Model
    elif model_type == "vlmap_answer_noc":
        from .model_vlmap_answer_noc import Model
    elif model_type == "vlmap_answer_nocarch":
        from .model_vlmap_answer_nocarch import Model
    elif model_type in ("vlmap_answer_vqa_all", "vlmap_answer_"):        # the reference's importer accepts both spellings (:33)
        from .model_vlmap_answer_vqa_all import Model
    elif model_type == "vlmap_answer_vqa_all2":
        from .model_vlmap_answer_vqa_all2 import Model
    elif model_type == "vlmap_answer2":
        from .model_vlmap_answer2 import Model
    elif model_type == "vlmap_answer_ent":
        from .model_vlmap_answer_ent import Model
    elif model_type == "vlmap_answer_adapt":
        from .model_vlmap_answer_adapt import Model
    elif model_type == "vlmap_answer_full":
        from .model_vlmap_answer_full import Model
    elif model_type == "vlmap_answer_no_noise":
        from .model_vlmap_answer_no_noise import Model
    elif model_type == "vqa":
        from .model_vqa import Model
    elif model_type == "vlmap_finetune":
        from .model_vlmap_finetune import Model
    elif model_type == "vlmap_only":
        from .model_vlmap_only import Model
    elif model_type in _REFERENCE_ONLY:
        raise ValueError("model_type %r is an ablation variant of the reference that is out of scope of the "
                         "MI355X hot path (supported: %s)" % (model_type, ", ".join(_NATIVE)))
    else:
        raise ValueError("Unknown model_type: {}".format(model_type))
    return Model
