"""config.yaml loader -- same keys as the reference (config.py:6-12, config.yaml.example:1-151),
attribute access like its EasyDict (easydict itself is not installed here)."""
import yaml


class Cfg(dict):
    """Minimal EasyDict stand-in: nested dicts become attribute-accessible."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Cfg):
            v = Cfg(v)
        elif isinstance(v, list):
            v = [Cfg(x) if isinstance(x, dict) else x for x in v]
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


DEFAULTS = dict(NUM_BBOXES_PER_CELL=5, MAX_NUM_BBOXES=13, LOCATION_LOSS_ALPHA=1000.0, BATCH_SIZE=32, INPUT_SIZE=299,
                NUM_TRAIN_EXAMPLES=56945, NUM_TRAIN_ITERATIONS=1000000, INITIAL_LEARNING_RATE=0.01, NUM_EPOCHS_PER_DELAY=4,
                LEARNING_RATE_DECAY_FACTOR=0.94, LEARNING_RATE_STAIRCASE=True, RMSPROP_DECAY=0.9, RMSPROP_MOMENTUM=0,
                RMSPROP_EPSILON=1.0, BATCHNORM_MOVING_AVERAGE_DECAY=0.9997, MOVING_AVERAGE_DECAY=0.9999,
                LOG_EVERY_N_STEPS=10, SAVE_INTERVAL_SECS=3600, MAX_TO_KEEP=3)


def parse_config_file(path_to_config):
    """config.py:6-12 (safe_load instead of the reference's Loader-less yaml.load)."""
    with open(path_to_config) as f:
        cfg = yaml.safe_load(f)
    return Cfg(cfg)


def with_defaults(cfg):
    out = Cfg(DEFAULTS)
    for k, v in cfg.items():
        out[k] = v
    return out
