"""minimal stand-in: see ../README.md"""
