"""Stand-in for `yolox.exp` of the un-vendored `yolox==0.3` package (test infrastructure only; see ../../README.md).

The reference's cfgs (`/root/reference/cfgs/*.py:7`: `from yolox.exp import Exp as MyExp`) subclass `Exp`, call
`super().__init__()` and then overwrite attributes; `tools/train.py:123` / `tools/eval.py:203` locate the class through
`get_exp(exp_file, name)` and customise it with `exp.merge(opts)`.  Only that surface is restated here, with the
defaults listed in SURVEY.md Appendix B — enough for `Exp().get_model()` (the part on the hot path) to run unchanged."""
import ast
import importlib.util
import os
import sys


class BaseExp:
    def __init__(self):
        self.seed = None
        self.output_dir = "./YOLOX_outputs"
        self.print_interval = 100
        self.eval_interval = 10

    def merge(self, cfg_list):
        """`-o key value key value ...` of tools/train.py: values are literal-evaluated into the attribute's type."""
        assert len(cfg_list) % 2 == 0
        for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
            if not hasattr(self, k):
                continue
            old = getattr(self, k)
            if old is not None and not isinstance(v, type(old)):
                try:
                    v = type(old)(v)
                except Exception:
                    v = ast.literal_eval(v)
            setattr(self, k, v)


class Exp(BaseExp):
    def __init__(self):
        super().__init__()
        self.num_classes = 80
        self.depth, self.width, self.act = 1.00, 1.00, "silu"
        self.data_num_workers = 4
        self.input_size, self.test_size = (640, 640), (640, 640)
        self.multiscale_range = 5
        self.data_dir, self.train_ann, self.val_ann = None, "instances_train2017.json", "instances_val2017.json"
        self.mosaic_prob = self.mixup_prob = 1.0
        self.degrees, self.translate, self.mosaic_scale, self.shear = 10.0, 0.1, (0.1, 2), 2.0
        self.enable_mixup = True
        self.warmup_epochs, self.max_epoch, self.no_aug_epochs = 5, 300, 15
        self.warmup_lr, self.basic_lr_per_img, self.min_lr_ratio = 0, 0.01 / 64.0, 0.05
        self.scheduler = "yoloxwarmcos"
        self.ema, self.weight_decay, self.momentum = True, 5e-4, 0.9
        self.print_interval, self.eval_interval = 10, 10
        self.save_history_ckpt = True
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.test_conf, self.nmsthre = 0.01, 0.65


def get_exp(exp_file=None, exp_name=None):
    """yolox.exp.get_exp for the `-f <file>` form: import the cfg file as a module and instantiate its `Exp`."""
    assert exp_file is not None, "the stand-in resolves experiment FILES only"
    sys.path.append(os.path.dirname(exp_file))
    spec = importlib.util.spec_from_file_location(os.path.basename(exp_file).split(".")[0], exp_file)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Exp()
