"""vqa/model_vlmap_answer_nocarch.py is byte for byte vqa/model_vlmap_answer_noc.py; so is this class."""
from .model_vlmap_answer_noc import Model as _Noc


class Model(_Noc):
    MODEL_TYPE = "vlmap_answer_nocarch"
