from .discriminator import Discriminator as UnivNetDiscriminator
from .generator import Generator as HifiGANGenerator
