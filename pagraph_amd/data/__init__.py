from .get_data import *  # noqa: F401,F403
