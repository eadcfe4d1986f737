"""Internal package of the MI355X-native Depth-VO-Feat training path: ctypes binding (lib), autograd
wrappers (ops, conv), device arenas and the data-parallel engine."""
