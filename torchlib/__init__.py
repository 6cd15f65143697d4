"""Import shim: the names PriMIA's scripts and checkpoints reach through `torchlib.*`, served by primia_amd.

A reference checkpoint pickles its `torchlib.utils.Arguments` instance (torchlib/utils.py:1489); with this
package on the path such a file unpickles against the HIP-backed classes, and `from torchlib.utils import
Arguments, LearningRateScheduler, MixUp, train, test, ...` in a reference-style script resolves here.  There
is no arithmetic in this package."""
