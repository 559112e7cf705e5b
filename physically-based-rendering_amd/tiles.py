"""Tile sharding of the framebuffer across GPUs — layout arithmetic shared by
bench.py and the tests.  The reference is single-device; pixels are independent given the frame
seed (pathtracing.cl:28,255,332), so 8x8-pixel tiles are dealt round-robin along a dealing order — row-major
with row ty rotated by DEAL_SHIFT * ty columns (csrc/pt_kernel.hpp: plain row-major order hands whole tile columns to
a rank whenever tiles_x is a multiple of world, and columns do not cost the same):

    position p( tx, ty ) = ty * tiles_x + ( tx + DEAL_SHIFT * ty ) % tiles_x
    owner( tile ) = p % world            local index of the tile on its owner = p // world        (world = 1: p = t)

Each rank keeps its tiles in a compact tile-major buffer (64 pixels x RGBA32F = 1 KiB per
tile, lane = (y % 8) * 8 + x % 8), padded to ceil( tiles / world ) tiles, which is exactly what
one all-gather concatenates.  These numpy functions mirror the device kernels untile /
retile / scatterGathered (csrc/pt_kernel.hpp) and are what the GPU tests check them against.
"""
import numpy as np

TILE = 8


def tile_counts(width, height, world):
    tiles_x, tiles_y = width // TILE, height // TILE
    total = tiles_x * tiles_y
    return tiles_x, tiles_y, total, (total + world - 1) // world


DEAL_SHIFT = 5


def deal_order(width, height, world):
    """tile at every position of the dealing order."""
    tiles_x, tiles_y, total, _ = tile_counts(width, height, world)
    p = np.arange(total)
    if world <= 1:
        return p
    ty, shifted = p // tiles_x, p % tiles_x
    return ty * tiles_x + (shifted - DEAL_SHIFT * ty) % tiles_x


def local_tile_ids(width, height, world, rank):
    """Global tile index of this rank's local tiles 0, 1, 2, ..."""
    return deal_order(width, height, world)[rank::world]


def to_tile_major(image):
    """(H, W, 4) row-major -> (tiles, 64, 4), tile t = (y // 8) * tiles_x + x // 8."""
    h, w, c = image.shape
    t = image.reshape(h // TILE, TILE, w // TILE, TILE, c).transpose(0, 2, 1, 3, 4)
    return t.reshape(-1, TILE * TILE, c)


def from_tile_major(tiles, width, height):
    c = tiles.shape[-1]
    t = tiles.reshape(height // TILE, width // TILE, TILE, TILE, c).transpose(0, 2, 1, 3, 4)
    return t.reshape(height, width, c)


def pack_rank_tiles(image, world, rank):
    """This rank's compact buffer (per_rank, 64, 4) from a full row-major image (zero padded)."""
    h, w, _ = image.shape
    _, _, _, per_rank = tile_counts(w, h, world)
    mine = to_tile_major(image)[local_tile_ids(w, h, world, rank)]
    out = np.zeros((per_rank,) + mine.shape[1:], image.dtype)
    out[:len(mine)] = mine
    return out


def unpack_gathered(gathered, width, height, world):
    """(world, per_rank, 64, 4) all-gather result -> (H, W, 4) row-major full frame."""
    _, _, total, per_rank = tile_counts(width, height, world)
    gathered = np.asarray(gathered).reshape(world, per_rank, TILE * TILE, -1)
    tiles = np.empty((total,) + gathered.shape[2:], gathered.dtype)
    for rank in range(world):
        ids = local_tile_ids(width, height, world, rank)
        tiles[ids] = gathered[rank, :len(ids)]
    return from_tile_major(tiles, width, height)


def rows_of_rank(width, height, world, rank):
    """Mask (H, W) of the pixels rank owns."""
    tiles_x, tiles_y, total, _ = tile_counts(width, height, world)
    owner = np.empty(total, np.int64)
    owner[deal_order(width, height, world)] = np.arange(total) % world
    owner = owner.reshape(tiles_y, tiles_x)
    return np.kron(owner == rank, np.ones((TILE, TILE), bool))
