"""rqvae/models/interfaces.py:20-72 of the reference (abstract stage-1 / stage-2 interfaces)."""
import abc

from torch import nn


class Stage1Model(nn.Module, metaclass=abc.ABCMeta):

    @abc.abstractmethod
    def get_codes(self, *args, **kwargs):
        """Generate the code from the input."""

    @abc.abstractmethod
    def decode_code(self, *args, **kwargs):
        """Generate the decoded image from the given code."""

    @abc.abstractmethod
    def get_recon_imgs(self, *args, **kwargs):
        """Scales the real and recon images properly."""

    @abc.abstractmethod
    def compute_loss(self, *args, **kwargs):
        """Compute the losses necessary for training."""


class Stage2Model(nn.Module, metaclass=abc.ABCMeta):

    @abc.abstractmethod
    def compute_loss(self, *args, **kwargs):
        """Compute the losses necessary for training."""

    def get_block_size(self):
        return self.block_size
