"""models package of the reference (/root/reference/models/__init__.py:5-44): dynamic lookup
`models.<name>_model.<Name>Model`, create_model(opt), get_options_modifier(name)."""
import importlib

from .base_model import BaseModel  # noqa: F401


def find_model_using_name(model_name):
    model_filename = __name__ + "." + model_name + "_model"
    try:
        modellib = importlib.import_module(model_filename)
    except ImportError as e:
        raise NotImplementedError("model [%s] is not implemented in swapnet_amd (warp | texture)" % model_name) from e
    target = model_name.replace("_", "") + "model"
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            return cls
    raise NotImplementedError("In %s.py, there should be a subclass of BaseModel with class name that matches %s "
                              "in lowercase." % (model_filename, target))


def get_options_modifier(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    model = find_model_using_name(opt.model)
    instance = model(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
