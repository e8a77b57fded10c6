import importlib

registry = {}


def register(id, entry_point, max_episode_steps=None, **kw):
    registry[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=kw)


def make(id, **kwargs):
    spec = registry[id]
    mod, cls = spec['entry_point'].split(':')
    env = getattr(importlib.import_module(mod), cls)(**kwargs)
    env.spec = spec
    return env
