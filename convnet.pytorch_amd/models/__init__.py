"""Model registry: every lowercase callable exported here is a ``--model`` choice, exactly like
/root/reference models/__init__.py:1-13 + main.py:24-26 (``models.__dict__[name](**config)``).
Only the families on the hot path are registered natively (resnet, mnist); the registry mechanism
itself is unchanged so further families can be added without touching the engine."""
from .resnet import *   # noqa: F401,F403
from .mnist import *    # noqa: F401,F403
