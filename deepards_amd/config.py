"""Configuration merge of the reference's ``deepards/config.py:6-22``: values come from, in rising precedence,

    1. ``defaults.yml`` next to this file,
    2. the override file given with ``-co / --config-override`` (an experiment file),
    3. the command line (every parser default is None, so only flags the user really gave win).

``args.<name>`` reads the merged dict (``__getattr__``), exactly how ``train_ards_detector.py`` consumes it.  A CLI
key that neither file knows is kept with its None / parser default (config.py:17-19), so every parser destination is
readable.  Below the two files sit the build's own knobs (``BUILD_DEFAULTS``: device stores handed in by a caller,
hipGraph switch, seed), which the reference does not have.  YAML is read with ``yaml.safe_load`` (nothing but plain
scalars / lists is expected in these files).
"""
import os

import yaml

DEFAULTS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'defaults.yml')


class Configuration(object):
    def __init__(self, parser_args, build_defaults=None):
        conf = dict(build_defaults or {})
        with open(DEFAULTS_FILE) as defaults:
            conf.update(yaml.safe_load(defaults) or {})
        override = getattr(parser_args, 'config_override', None)
        if override:
            with open(override) as overrides_f:
                conf.update(yaml.safe_load(overrides_f) or {})
        for k, v in vars(parser_args).items():
            if v is not None or k not in conf:
                conf[k] = v
        self.__dict__['conf'] = conf

    def __getattr__(self, attr):
        try:
            return self.__dict__['conf'][attr]
        except KeyError:
            raise AttributeError(attr)

    def __setattr__(self, attr, value):          # ``args.network = ...`` in main() writes through to the dict
        self.__dict__['conf'][attr] = value

    def __contains__(self, attr):
        return attr in self.__dict__['conf']
