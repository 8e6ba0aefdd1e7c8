"""TEST STUB (tests/stubs/README.md): rqvae/utils/config.py:10-21 of the reference only isinstance-checks EasyDict."""


class EasyDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
