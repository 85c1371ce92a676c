"""Config helpers of the drop-in surface (ppvector/utils/utils.py:32-52, :78-83)."""
import numpy as np


def strtobool(v):
    v = str(v).lower()
    if v in ('y', 'yes', 't', 'true', 'on', '1'):
        return 1
    if v in ('n', 'no', 'f', 'false', 'off', '0'):
        return 0
    raise ValueError(f'invalid truth value {v!r}')


def add_arguments(argname, type, default, help, argparser, **kwargs):
    type = strtobool if type == bool else type
    argparser.add_argument("--" + argname, default=default, type=type, help=help + ' 默认: %(default)s.', **kwargs)


class Dict(dict):
    __setattr__ = dict.__setitem__
    __getattr__ = dict.__getitem__


def dict_to_object(dict_obj):
    if not isinstance(dict_obj, dict):
        return dict_obj
    inst = Dict()
    for k, v in dict_obj.items():
        inst[k] = dict_to_object(v)
    return inst


def cosin_metric(x1, x2):
    return np.dot(x1, x2) / (np.linalg.norm(x1) * np.linalg.norm(x2))
