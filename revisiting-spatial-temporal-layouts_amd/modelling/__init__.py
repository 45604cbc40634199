"""Drop-in counterparts of the reference's ``src/modelling`` for the STLT path (configs + models)."""
