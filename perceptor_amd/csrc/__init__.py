"""csrc package of perceptor_amd (MI355X-native guided-diffusion hot path)."""
