"""Masked loss and accuracy of the reference (gcn/metrics.py:85-101) on GPU tensors."""
import torch


def masked_softmax_cross_entropy(preds, labels, mask):
    """Softmax cross-entropy with masking: mean over ALL rows of CE_i * mask_i / mean(mask)
    (gcn/metrics.py:85-92).  ``labels`` may hold all-zero rows (CiteSeer's padded nodes): their
    cross-entropy is 0, as tf.nn.softmax_cross_entropy_with_logits gives."""
    labels = labels.to(torch.float32)
    loss = -(labels * torch.log_softmax(preds, dim=1)).sum(dim=1)
    mask = mask.to(torch.float32)
    mask = mask / mask.mean()
    return (loss * mask).mean()


def masked_accuracy(preds, labels, mask):
    """Accuracy with masking (gcn/metrics.py:94-101); argmax of an all-zero label row is 0."""
    correct = (torch.argmax(preds, 1) == torch.argmax(labels, 1)).to(torch.float32)
    mask = mask.to(torch.float32)
    mask = mask / mask.mean()
    return (correct * mask).mean()
