"""MI355X counterpart of vqa/model_standard.py: the same network trained from scratch
(every variable trainable, fusion MLP under 'reasoning/', plain Xavier 'classifier'
head, loss without the train-answer mask -- vqa/model_standard.py:80-84, 251-285)."""
from .model_vlmap_answer import Model as _Base


class Model(_Base):
    MODEL_TYPE = "standard"
