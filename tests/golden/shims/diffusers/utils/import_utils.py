def is_xformers_available():
    return False
