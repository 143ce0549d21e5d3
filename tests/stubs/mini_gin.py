"""A FUNCTIONAL stand-in for gin-config, test infrastructure only (gin is not installed in this image).  It implements
the part of gin's behaviour the reference relies on (internal/configs.py:11-22,186-187; internal/models.py:22,30,688,693;
train.py:41): `Name.param = literal` bindings parsed from .gin files / binding strings and injected as keyword arguments
when a `@gin.configurable` class is constructed (class mutated in place: its construction `__init__` is wrapped, as gin
does for classes) -- only for parameters the callee accepts, any name when it takes **kwargs; explicit arguments win."""
import ast
import functools
import inspect
import os
import types

REQUIRED = object()
_BINDINGS = {}        # (configurable name, parameter) -> value
_REGISTRY = {}
_SEARCH = ['']


def add_config_file_search_path(p):
    _SEARCH.append(p)


def clear_config():
    _BINDINGS.clear()


def _accepts(fn):
    sig = inspect.signature(fn)
    names = {n for n, p in sig.parameters.items() if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}
    var_kw = any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
    return names, var_kw


def _inject(name, fn, kwargs):
    names, var_kw = _accepts(fn)
    for (n, param), v in _BINDINGS.items():
        if n == name and param not in kwargs and (var_kw or param in names):
            kwargs[param] = v
    return kwargs


def _decorate(obj, name):
    name = name or obj.__name__
    if name in _REGISTRY and _REGISTRY[name] is not obj:
        raise ValueError(f"A configurable matching '{name}' already exists.")
    _REGISTRY[name] = obj
    if isinstance(obj, type):
        orig = obj.__init__

        @functools.wraps(orig)
        def init(self, *a, **k):
            orig(self, *a, **_inject(name, orig, dict(k)))
        obj.__init__ = init
        return obj

    @functools.wraps(obj)
    def wrapper(*a, **k):
        return obj(*a, **_inject(name, obj, dict(k)))
    return wrapper


def configurable(name_or_fn=None, module=None, allowlist=None, denylist=None, **_):
    if callable(name_or_fn):
        return _decorate(name_or_fn, None)
    return lambda f: _decorate(f, name_or_fn)


def _external_configurable(fn, name=None, module=None, **_):
    return fn


config = types.SimpleNamespace(external_configurable=_external_configurable)
external_configurable = _external_configurable


def parse_config(text, skip_unknown=False):
    if isinstance(text, (list, tuple)):
        text = "\n".join(text)
    for line in text.splitlines():
        line = line.split('#', 1)[0].strip() if not ("'" in line or '"' in line) else line.strip()
        if not line or line.startswith('#') or '=' not in line:
            continue
        lhs, rhs = line.split('=', 1)
        lhs, rhs = lhs.strip(), rhs.strip()
        target, param = lhs.rsplit('.', 1)
        target = target.rsplit('/', 1)[-1].rsplit('.', 1)[-1]          # scopes / module prefixes are not modelled
        try:
            val = ast.literal_eval(rhs)
        except (ValueError, SyntaxError):
            val = rhs
        _BINDINGS[(target, param)] = val


def parse_config_files_and_bindings(config_files, bindings, skip_unknown=False, **_):
    for f in config_files or []:
        for d in _SEARCH:
            p = os.path.join(d, f)
            if os.path.isfile(p):
                parse_config(open(p).read(), skip_unknown)
                break
        else:
            raise IOError(f"Unable to open file: {f}")
    parse_config(list(bindings or []), skip_unknown)


def config_str():
    return "".join(f"{n}.{p} = {v!r}\n" for (n, p), v in sorted(_BINDINGS.items()))


class config_scope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
