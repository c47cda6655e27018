class AttnProcessor:          # only used in type annotations by the reference
    pass
