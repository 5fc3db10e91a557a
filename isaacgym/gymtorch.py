"""isaacgym.gymtorch: tensors of this framework are torch tensors already."""


def wrap_tensor(t):
    return t


def unwrap_tensor(t):
    return t
