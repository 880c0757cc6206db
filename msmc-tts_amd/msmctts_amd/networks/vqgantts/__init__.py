from .msmc_vqgan import MSMCVQGAN
