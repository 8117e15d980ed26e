"""The slice of the reference's `diffusion/logger.py` the training loop touches (:212-244, :325-373):
per-iteration key/value store with running means.  Output formats (csv/tensorboard/json writers) are the
train platform's business and out of scope."""
import os
import tempfile
from collections import defaultdict

_CURRENT = None


class Logger:
    def __init__(self, dir):
        self.dir = dir
        self.name2val = defaultdict(float)
        self.name2cnt = defaultdict(int)

    def logkv(self, key, val):
        self.name2val[key] = val

    def logkv_mean(self, key, val):
        oldval, cnt = self.name2val[key], self.name2cnt[key]
        self.name2val[key] = oldval * cnt / (cnt + 1) + val / (cnt + 1)
        self.name2cnt[key] = cnt + 1

    def dumpkvs(self):
        out = dict(self.name2val)
        self.name2val.clear()
        self.name2cnt.clear()
        return out

    def get_dir(self):
        return self.dir


def configure(dir=None, format_strs=None, comm=None, log_suffix=""):
    global _CURRENT
    if dir is None:
        dir = os.getenv("OPENAI_LOGDIR") or tempfile.mkdtemp(prefix="mst_log_")
    os.makedirs(os.path.expanduser(dir), exist_ok=True)
    _CURRENT = Logger(os.path.expanduser(dir))
    return _CURRENT


def get_current():
    if _CURRENT is None:
        configure()
    return _CURRENT


def logkv(key, val):
    get_current().logkv(key, val)


def logkv_mean(key, val):
    get_current().logkv_mean(key, val)


def dumpkvs():
    return get_current().dumpkvs()


def getkvs():
    return get_current().name2val


def get_dir():
    return get_current().get_dir()


def log(*args, **kw):
    print(*args)
