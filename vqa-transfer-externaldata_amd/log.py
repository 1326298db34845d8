"""Logger with the reference's INFOV level (util/util.py:9-46) minus the colorlog dependency."""
import logging

log = logging.getLogger("vqa_hot")
if not log.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("[%(asctime)s] %(message)s"))
    log.addHandler(_h)
    log.setLevel(logging.INFO)
    log.propagate = False
logging.addLevelName(logging.INFO + 1, "INFOV")


def _infov(msg, *args, **kwargs):
    log.log(logging.INFO + 1, msg, *args, **kwargs)


log.infov = _infov
