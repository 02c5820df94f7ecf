from .sam import Sam
from .image_encoder import ImageEncoderViT
from .pos_encoder import PromptEncoder
from .transformer import TwoWayTransformer
from .box_decoder import MaskDecoder
