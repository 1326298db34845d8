"""MI355X counterpart of vqa/model_standard_testmask.py: the earlier form of model_standard.

Same network as model_standard (every variable trainable: :64-68; transfer variables 'encode_L' / 'GloVe': :70-77);
the training loss is the sigmoid cross-entropy masked by the train-answer mask while the reported loss is not
(:262-268), and `report` carries the nine scalars of :295-304 under their older names.  The reference class predates
the `output` / `heavy_output` dictionaries and the `image_features` argument; both are accepted / filled here as in
the other models (a superset: the Trainer only reads `loss`, `report` and the variable filters)."""
from .model_vlmap_answer import Model as _Base

# vqa/model_standard_testmask.py:295-304 in terms of the step's 13 report scalars
REPORT_KEYS = (("answer_train_loss", "answer_train_loss"), ("answer_report_loss", "answer_report_loss"),
               ("answer_accuracy", "answer_acc"), ("exist_answer_accuracy", "exist_acc"),
               ("test_answer_accuracy", "test_acc"), ("normal_test_answer_accuracy", "normal_test_acc"),
               ("max_exist_answer_accuracy", "max_exist_acc"), ("test_max_answer_accuracy", "test_max_acc"),
               ("test_max_exist_answer_accuracy", "test_max_exist_acc"))


class Model(_Base):
    MODEL_TYPE = "standard_testmask"
    REPORT_RENAME = REPORT_KEYS
