from .msmc_vqgan import MSMCVQGAN
from .msmc_vqgan_emb import MSMCVQGANEmb
