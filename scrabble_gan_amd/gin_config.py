"""A reader for the subset of gin-config syntax the reference uses
(/root/reference/src/scrabble_gan.gin, parsed by gin.parse_config_file at main.py:56):
`scope.param = value`, '#' comments (also trailing), floats like 2E-4, ints, tuples, quoted strings,
None/True/False and `@name` references to registered configurables.  gin-config itself is not
installed in this image."""
from __future__ import annotations

import ast
import functools
import inspect
import re
from typing import Any, Callable, Dict

_BINDINGS: Dict[str, Dict[str, Any]] = {}
_REGISTRY: Dict[str, Callable] = {}


class GinError(ValueError):
    pass


def clear_config():
    _BINDINGS.clear()


def external_configurable(fn: Callable, name: str = None) -> Callable:
    """gin.external_configurable (main.py:16-18): make `fn` addressable as @name."""
    _REGISTRY[name or fn.__name__] = fn
    return fn


def _strip_comment(line: str) -> str:
    out, quote = [], None
    for ch in line:
        if quote:
            out.append(ch)
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
            out.append(ch)
        elif ch == "#":
            break
        else:
            out.append(ch)
    return "".join(out).strip()


def _parse_value(text: str) -> Any:
    text = text.strip()
    if text.startswith("@"):
        name = text[1:].rstrip("()")
        if name not in _REGISTRY:
            raise GinError("unknown configurable reference @%s" % name)
        return _REGISTRY[name]
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError) as e:
        raise GinError("cannot parse value %r" % text) from e


def parse_config(text: str) -> None:
    for ln, raw in enumerate(text.splitlines(), 1):
        line = _strip_comment(raw)
        if not line:
            continue
        m = re.match(r"^([A-Za-z_][\w/]*)\.([A-Za-z_]\w*)\s*=\s*(.+)$", line)
        if not m:
            raise GinError("line %d: expected `scope.param = value`, got %r" % (ln, raw))
        scope, param, value = m.group(1), m.group(2), _parse_value(m.group(3))
        _BINDINGS.setdefault(scope, {})[param] = value


def parse_config_file(path: str) -> None:
    with open(path) as f:
        parse_config(f.read())


def bind_parameter(key: str, value: Any) -> None:
    scope, param = key.rsplit(".", 1)
    _BINDINGS.setdefault(scope, {})[param] = value


def query_parameter(key: str) -> Any:
    scope, param = key.rsplit(".", 1)
    return _BINDINGS[scope][param]


def configurable(name_or_fn=None):
    """@gin.configurable / @gin.configurable('scope') (main.py:25,38,43): parameters the caller does
    not pass are taken from the parsed bindings of the scope."""
    def wrap(fn, scope):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def inner(*args, **kwargs):
            bound = sig.bind_partial(*args, **kwargs)
            for k, v in _BINDINGS.get(scope, {}).items():
                if k not in sig.parameters:
                    raise GinError("%s has no parameter %r" % (scope, k))
                if k not in bound.arguments:
                    kwargs[k] = v
            missing = [p for p in sig.parameters.values()
                       if p.default is inspect.Parameter.empty and p.name not in bound.arguments and p.name not in kwargs
                       and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
            if missing:
                raise GinError("no value for %s.%s (not passed and not bound in the gin config)" % (scope, missing[0].name))
            return fn(*args, **kwargs)
        _REGISTRY[scope] = inner
        return inner
    if callable(name_or_fn):
        return wrap(name_or_fn, name_or_fn.__name__)
    return lambda fn: wrap(fn, name_or_fn or fn.__name__)
