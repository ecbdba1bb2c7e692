from .cross_entropy import (SoftTargetCrossEntropy, TokenLabelGTCrossEntropy,  # noqa: F401
                            TokenLabelSoftTargetCrossEntropy, TokenLabelCrossEntropy, SparseTokenLabelTarget)
