mkdir -p gpurun_out/r6
for sc in sponza dragon; do for t in eight-order eight-order-compact; do
  python bench.py --scene $sc --traversal $t --cpu-seconds 0 --hold-seconds 0 --modes off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-20s %-12s %8.1f Msamples/s  %.4f ms/step  walk bytes %d' % (d['config']['scene'], d['config']['traversal'], d['schedule'], d['value'], d['ms_per_step'], d['config']['scene_device_bytes']['walk_streams']))"
done; done
for t in eight-order eight-order-compact; do
  python bench.py --scene hairball --width 3840 --height 2160 --steps 16 --warmup 16 --traversal $t --cpu-seconds 0 --hold-seconds 0 --modes off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-20s %-12s %8.1f Msamples/s  %.4f ms/step  walk bytes %d' % (d['config']['scene']+'4k', d['config']['traversal'], d['schedule'], d['value'], d['ms_per_step'], d['config']['scene_device_bytes']['walk_streams']))"
done
