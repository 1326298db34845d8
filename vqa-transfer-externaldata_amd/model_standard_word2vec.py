"""MI355X counterpart of vqa/model_standard_word2vec.py: model_standard with a word-embedding answer head.

Differences from model_standard (vqa/model_standard_word2vec.py:180-202, 278-280): the 'classifier' fc_layer maps the
joint feature to 300 dimensions and the logits are its product with a FIXED [300, A] matrix holding the GloVe vectors
of the answers (modules.LearnGloVe(answer_dict, learnable=False, oov_mean_initialize=True): multi-word answers take
the mean of their words' vectors); the training loss is masked by the train-answer mask, the reported loss is not.
The matrix is a tf.constant in the reference, so it is neither trained nor saved in checkpoints."""
import numpy as np

from .model_vlmap_answer import Model as _Base, learn_glove_init


class Model(_Base):
    MODEL_TYPE = "standard_word2vec"

    def _engine_kwargs(self):
        cfg = self.config
        glove = getattr(cfg, "glove", None)
        if glove is None and not (getattr(cfg, "debug", 0) or getattr(cfg, "synthetic", 0)):
            raise ValueError("model_standard_word2vec needs the GloVe vectors of the answers (config.glove = "
                             "{'dict': word -> row, 'param': [n, 300]}): data/preprocessed/glove.6B.300d.hdf5 and "
                             "glove_vocab.json are download-only")
        rng = np.random.default_rng(int(getattr(cfg, "seed", 123)) + 1) if glove is None else None
        w = learn_glove_init(self.answer_dict, glove, rng, oov_mean_initialize=True)        # [A, 300]
        if glove is None:
            w = w * 30.0            # synthetic stand-in at the scale of GloVe vectors (~0.3) rather than 0.01
        return {"answer_glove": np.ascontiguousarray(w.T)}
