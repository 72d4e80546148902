from .resize import resize, resize_backward
