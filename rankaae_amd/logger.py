"""File/stream loggers with the reference's two formats (``sc/utils/logger.py:5-34``):
timestamped messages, or bare ``%(message)s`` lines for ``losses.csv``."""
import logging
import os


def create_logger(logger_name, log_path=None, append=False, simple_fmt=False):
    if log_path is not None and not append and os.path.isfile(log_path):
        open(log_path, "w").close()
    logger = logging.getLogger(logger_name)
    logger.setLevel(logging.DEBUG)
    handler = logging.StreamHandler() if log_path is None else logging.FileHandler(log_path)
    handler.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(message)s") if simple_fmt else \
        logging.Formatter("%(asctime)s %(levelname)s:  %(message)s", datefmt="%m-%d %H:%M")
    handler.setFormatter(fmt)
    logger.addHandler(handler)
    return logger
