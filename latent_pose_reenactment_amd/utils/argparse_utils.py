"""Argument-parser helpers with the reference's surface (utils/argparse_utils.py:4-46): ``parser.add(...)`` and the
``store_bool`` action that turns ``--flag`` into a ``--flag`` / ``--no-flag`` pair."""
import argparse


class _BoolPair(argparse.Action):
    """``action='store_bool'``: registers both ``--name`` (True) and ``--no-name`` (False) for one destination."""

    def __init__(self, option_strings, dest, default=None, required=False, help='', **_ignored):
        if len(option_strings) != 1 or not option_strings[0].startswith('--'):
            raise ValueError('store_bool needs exactly one long option')
        stem = option_strings[0][2:]
        super().__init__([f'--{stem}', f'--no-{stem}'], dest=dest, nargs=0, default=default, required=required,
                         help=f'{help} (--{stem} / --no-{stem})')

    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, not option_string.startswith('--no-'))


class MyArgumentParser(argparse.ArgumentParser):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.register('action', 'store_bool', _BoolPair)

    add = argparse.ArgumentParser.add_argument
