"""Wire codec of the reference's protobuf message ``isg_ai.ImageYoloBoxesPair``
(isg_ai.proto:15-31, proto2, all fields optional) without the protobuf runtime.

field  name        type    wire
  1    channels    int32   varint
  2    img_height  int32   varint
  3    img_width   int32   varint
  4    image       bytes   length-delimited   (HWC, dtype = img_type)
  5    box_count   int32   varint
  6    boxes       bytes   length-delimited   ([n,5] int32 x,y,w,h,class; absent when n == 0, build_lmdb.py:62-63)
  7    img_type    string  length-delimited   (NumPy dtype.str, e.g. '|u1')
  8    box_type    string  length-delimited
  9    label       int32   varint             (unused by the reference)
Pinned by tests/golden/proto_pair.bin, serialised by the reference's own generated isg_ai_pb2 module.
"""
import numpy as np

_VARINT, _LEN = 0, 2
_FIELDS = {1: ('channels', _VARINT), 2: ('img_height', _VARINT), 3: ('img_width', _VARINT), 4: ('image', _LEN), 5: ('box_count', _VARINT),
           6: ('boxes', _LEN), 7: ('img_type', _LEN), 8: ('box_type', _LEN), 9: ('label', _VARINT)}


def _put_varint(out, v):
    if v < 0:
        v += 1 << 64            # int32 negatives are sign-extended to 10 bytes
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return


def _get_varint(buf, pos):
    shift = 0
    v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            break
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')
    return v, pos


class ImageYoloBoxesPair:
    """Same attribute names as the generated class; ParseFromString / SerializeToString like protobuf."""

    def __init__(self):
        self.channels = self.img_height = self.img_width = self.box_count = self.label = 0
        self.image = b''
        self.boxes = b''
        self.img_type = ''
        self.box_type = ''
        self._present = set()

    def SerializeToString(self):
        out = bytearray()
        for num in sorted(_FIELDS):
            name, wt = _FIELDS[num]
            if name not in self._present and not getattr(self, name):
                continue
            val = getattr(self, name)
            _put_varint(out, (num << 3) | wt)
            if wt == _VARINT:
                _put_varint(out, int(val))
            else:
                raw = val.encode('utf-8') if isinstance(val, str) else bytes(val)
                _put_varint(out, len(raw))
                out += raw
        return bytes(out)

    def ParseFromString(self, data):
        buf = memoryview(data)
        pos, n = 0, len(buf)
        self.__init__()
        while pos < n:
            tag, pos = _get_varint(buf, pos)
            num, wt = tag >> 3, tag & 7
            if wt == _VARINT:
                v, pos = _get_varint(buf, pos)
                val = v - (1 << 64) if v >= 1 << 63 else v
            elif wt == _LEN:
                ln, pos = _get_varint(buf, pos)
                val = bytes(buf[pos:pos + ln])
                pos += ln
            elif wt == 1:
                val = bytes(buf[pos:pos + 8])
                pos += 8
            elif wt == 5:
                val = bytes(buf[pos:pos + 4])
                pos += 4
            else:
                raise ValueError('unsupported wire type %d' % wt)
            if num in _FIELDS:
                name, _ = _FIELDS[num]
                if name in ('img_type', 'box_type'):
                    val = val.decode('utf-8')
                setattr(self, name, val)
                self._present.add(name)
        return self

    # ---- helpers mirroring build_lmdb.write_img_to_db / imagereader.__image_loader ----
    @staticmethod
    def from_arrays(img, boxes):
        """build_lmdb.py:46-69: img HWC ndarray, boxes [n,5] int32 or None."""
        m = ImageYoloBoxesPair()
        img = np.ascontiguousarray(img)
        m.channels = img.shape[2] if img.ndim == 3 else 1
        m.img_height, m.img_width = img.shape[0], img.shape[1]
        m.image = img.tobytes()
        m.img_type = img.dtype.str
        m._present |= {'channels', 'img_height', 'img_width', 'image', 'box_count', 'img_type', 'box_type'}
        if boxes is not None and len(boxes) > 0:
            boxes = np.ascontiguousarray(boxes, dtype=np.int32)
            m.box_count = boxes.shape[0]
            m.boxes = boxes.tobytes()
            m._present.add('boxes')
        else:
            m.box_count = 0
        m.box_type = np.dtype(np.int32).str
        return m

    def to_arrays(self):
        """imagereader.py:353-365 -> (img[H,W,C], boxes[n,5] int32)."""
        img = np.frombuffer(self.image, dtype=np.dtype(self.img_type)).reshape(self.img_height, self.img_width, self.channels)
        if self.box_count > 0:
            boxes = np.frombuffer(self.boxes, dtype=np.dtype(self.box_type)).reshape(self.box_count, 5)
        else:
            boxes = np.zeros((0, 5), np.int32)
        return img, boxes
