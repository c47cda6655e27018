"""tests/tools alias of the offline tokenizer"""
from e4t.utils import WhitespaceTokenizer as WordTokenizer  # noqa: F401
