"""One training step, as the reference's loop body runs it (trainer/trainer.py:114-136).

zero_grad -> forward -> criterion on every deep-supervision head -> mean over heads -> backward ->
optimizer.step().  The loss and the optimizer are PyTorch-ROCm host code; forward and backward of the
network are the HIP path.  Unlike the reference the head outputs are not moved to the CPU for the loss.
"""
from __future__ import annotations


def train_step(model, optimizer, criterion, inputs, target):
    optimizer.zero_grad()
    outputs = model(inputs)
    if isinstance(outputs, tuple):
        avgloss = 0
        for output in outputs:
            avgloss = avgloss + criterion(output, target)
        avgloss = 1.0 * avgloss / len(outputs)
    else:
        avgloss = criterion(outputs, target)
    avgloss.backward()
    optimizer.step()
    return outputs, avgloss
