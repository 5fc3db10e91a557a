"""Configuration data for the in-scope robots / envs / sensors / tasks.

Attribute names follow the reference's nested config classes (aerial_gym/config/**), so
user code that tweaks e.g. ``cfg.init_config.min_init_state`` or ``cfg.env.num_envs``
works unchanged; only the values the hot path consumes are carried.
"""
