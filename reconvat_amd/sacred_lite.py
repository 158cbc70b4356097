"""A tiny stand-in for the part of `sacred` the reference's training scripts use
(train_UNet_Onset_VAT.py:15,26-86): ``python script.py with key=value ...`` overrides of a config
function's local variables, values parsed as Python literals with a string fallback, unknown keys
rejected, entries injected into the main function by parameter name.  sacred itself is not installed on
the MI355X image; if it is, nothing here conflicts with it."""
import ast
import inspect
import sys


class ConfigError(KeyError):
    pass


def parse_cli(argv):
    """['with', 'a=1', 'b=foo'] -> {'a': 1, 'b': 'foo'} (also accepts the pairs without the 'with')."""
    out = {}
    args = list(argv)
    if args and args[0] == 'with':
        args = args[1:]
    for item in args:
        if '=' not in item:
            raise ConfigError(f"cannot parse '{item}': expected key=value")
        k, v = item.split('=', 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


class Experiment:
    def __init__(self, name):
        self.name = name
        self._config_fn = None
        self.config_values = {}

    def config(self, fn):
        self._config_fn = fn
        return fn

    def build_config(self, overrides):
        """Run the config function with overrides taking precedence over its own assignments (so derived
        entries such as `logdir` see the overridden values, like sacred's config scopes)."""
        base = self._config_fn({})
        unknown = [k for k in overrides if k not in base]
        if unknown:
            raise ConfigError(f'unknown config entries {unknown}; known: {sorted(base)}')
        cfg = self._config_fn(dict(overrides))
        cfg.update(overrides)
        self.config_values = cfg
        return cfg

    def run(self, main, argv=None):
        overrides = parse_cli(sys.argv[1:] if argv is None else argv)
        cfg = self.build_config(overrides)
        print('Configuration:')
        for k in sorted(cfg):
            mark = ' *' if k in overrides else ''
            print(f'  {k} = {cfg[k]!r}{mark}')
        names = inspect.signature(main).parameters
        return main(**{k: cfg[k] for k in names if k in cfg})

    def automain(self, main):
        if main.__module__ == '__main__':
            self.run(main)
        return main
