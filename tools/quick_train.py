import sys, time, yaml, torch
sys.path.insert(0,'.')
from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
from seqdex_amd.a2c_agent import A2CAgent
n=int(sys.argv[1]) if len(sys.argv)>1 else 1024
task_name=sys.argv[3] if len(sys.argv)>3 else 'BlockAssemblyGraspSim'
if task_name=='BlockAssemblyInsertSim':
    from seqdex_amd.tasks.block_assembly_insert_sim import BlockAssemblyInsertSim as BlockAssemblyGraspSim
elif task_name=='BlockAssemblyOrient':
    from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient as BlockAssemblyGraspSim
from seqdex_amd.config import TASK_CFG, TRAIN_CFG
cfg=yaml.safe_load(open('seqdex_amd/'+TASK_CFG[task_name])); cfg['env']['numEnvs']=n
tr=yaml.safe_load(open('seqdex_amd/'+TRAIN_CFG[task_name]))
if len(sys.argv)>4:
    tr['params']['config']['minibatch_size']=int(sys.argv[4]); tr['params']['config']['central_value_config']['minibatch_size']=int(sys.argv[4])
print('minibatch_size', tr['params']['config']['minibatch_size'])
t0=time.time()
task=BlockAssemblyGraspSim(cfg, device_type='cuda', device_id=0, headless=True, piles_per_type=4)
print('task create s', time.time()-t0)
env=RLgamesVecTaskPython(task,'cuda:0')
tr['params']['config'].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
agent=A2CAgent('run', tr['params'])
for ep in range(int(sys.argv[2]) if len(sys.argv)>2 else 4):
    r=agent.train_epoch()
    print('epoch',ep,'step %.4f play %.4f update %.4f total %.4f'%r[:4], 'a %.4f c %.4f kl %.5f lr %.2e'%(r[4][0],r[5][0],r[8][0],r[9]),
          'fps_step %.0f fps_total %.0f'%(n*8/r[0], n*8/r[3]), 'games', agent.game_rewards.get_mean(), agent.game_lengths.get_mean())
print('nc mean', task.sim.NCONTACTS.float().mean().item(), 'rew mean', task.rew_buf.mean().item())
